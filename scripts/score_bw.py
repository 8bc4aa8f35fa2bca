"""Diagnostic: streaming bandwidth of the score-table pass for build variants of score.hip
(-DPCG_SCORE_UNROLL / -DPCG_SCORE_BLOCKS_PER_CU / -DPCG_SCORE_NT), on a table far larger than the Infinity Cache."""
import ctypes as C, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pcgnn_amd import _lib
SRC = os.path.join(ROOT, "pc-gnn_amd", "csrc", "score.hip")
variants = [(8, 8, 1), (8, 8, 0), (4, 8, 1), (8, 16, 1), (8, 4, 1), (4, 16, 1)]
N, F = 8_000_000, 32
X = torch.randn(N, F)
W = torch.randn(2, F, device="cuda"); b = torch.zeros(2, device="cuda"); s0 = torch.empty(N, device="cuda")
import numpy as np
from pcgnn_amd.graph import DeviceGraph
g = DeviceGraph(X.cpu().numpy(), [(np.zeros(N + 1, dtype=np.int64), np.zeros(0, dtype=np.int32))], [], torch.device("cuda", 0))
del X
for unroll, bpc, nt in variants:
    so = f"/tmp/score_u{unroll}_b{bpc}_n{nt}.so"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", f"-DPCG_SCORE_UNROLL={unroll}",
                    f"-DPCG_SCORE_BLOCKS_PER_CU={bpc}", f"-DPCG_SCORE_NT={nt}", SRC, "-o", so], check=True,
                   stderr=subprocess.DEVNULL)
    lib = C.CDLL(so)
    fn = lib.pcg_score_table
    fn.restype = C.c_int
    args = (g.desc_ref(), C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), C.c_int64(0), C.c_int64(N),
            C.c_void_p(s0.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(3):
        assert fn(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(*args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"unroll {unroll} blocks/CU {bpc} nt {nt}: {ms*1e3:7.1f} us  {N*F*4/ms/1e6:7.1f} GB/s", flush=True)
