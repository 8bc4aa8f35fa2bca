"""Build libpcgnn_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m pcgnn_amd.build        # or:  __graft_entry__.build()
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libpcgnn_hip.so"
SOURCES = ["score.hip", "mark.hip", "sort.hip", "segmean_pick.hip", "choose.hip", "select.hip", "gather.hip", "dense.hip", "halo.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def lib_path() -> str:
    # PCG_LIB: a developer's A/B build of the same sources (scripts/ab_build.sh); unset = the in-tree library
    return os.environ.get("PCG_LIB") or os.path.join(LIBDIR, LIBNAME)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found - cannot build libpcgnn_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "choose.h"), os.path.join(HERE, "..", "include", "pcgnn.h")]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        objs.append(o)
    out = lib_path()
    if force or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build_library(verbose=True))
