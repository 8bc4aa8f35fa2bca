#!/usr/bin/env python3
"""Which kernels still have loads the compiler put behind a branch (+ a full vmcnt(0) wait)?  Reads the device assembly that
scripts/kernel_resources.sh leaves in /tmp/<file>.s.  A load inside `cond ? p[i] : 0` is compiled into a branch that waits
for every load in flight: gathers then go out one at a time.  Usage: python scripts/isa_load_scan.py [file.s ...]"""
import re
import sys

files = sys.argv[1:] or [f"/tmp/{n}.s" for n in ("choose", "select", "gather", "dense", "score", "halo", "segmean_pick", "sort")]
for path in files:
    try:
        txt = open(path).read()
    except OSError:
        continue
    for m in re.finditer(r"^(_ZN3pcg\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, lines = m.group(1), m.group(2).split("\n")
        is_load = lambda l: re.search(r"\b(global_load|buffer_load|flat_load)", l) is not None
        loads = sum(1 for l in lines if is_load(l))
        w0 = sum(1 for l in lines if "s_waitcnt vmcnt(0)" in l)
        guarded = 0
        for i, l in enumerate(lines):
            if "s_cbranch_execz" in l and any(is_load(x) for x in lines[i + 1:i + 7]):
                guarded += 1
        print(f"{name[:64]:64s} lines {len(lines):6d} loads {loads:4d} vmcnt(0) waits {w0:4d} loads right behind a branch {guarded:4d}")
