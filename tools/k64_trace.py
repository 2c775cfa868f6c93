#!/usr/bin/env python
"""Per-tile timeline of k_gemm_split_k64 from a DS2_K64_TRACE build (python tools/ab.py build k64trace -DDS2_K64_TRACE=1):
   DS2_LIB=.../ab_k64trace.so python tools/k64_trace.py [M]   (key projection of the memory attention: N=256, K=64, RoPE,
   hi plane only) - drives ds2_memory_attention's k_proj through bench-like shapes is overkill; a plain op_gemm shows the
   MFMA / slab timing without RoPE"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from det_sam2_amd.hip_model import HipOps
from det_sam2_amd import _capi
ops = HipOps("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 459776
A = torch.randn(M, 64, device="cuda"); W = torch.randn(256, 64, device="cuda") * 0.05; b = torch.randn(256, device="cuda")
for _ in range(3):
    ops.op_gemm(A, W, b, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.op_gemm(A, W, b, 0)
e1.record(); torch.cuda.synchronize()
print("op_gemm (incl. A split pre-pass) us:", e0.elapsed_time(e1) * 100)
lib = ctypes.CDLL(_capi.LIB_PATH)
buf = np.zeros((8, 256), dtype=np.uint64)
assert lib.ds2_debug_k64_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
per = 2 + 2 * 4
for w in (0, 1, 4):
    n = int((t[w] > 0).sum()) // per
    x = t[w, :n * per].reshape(n, per)
    print(f"wave {w}: tiles {n}")
    for i in range(n):
        d = np.diff(x[i])
        nxt = (t[w, (i + 1) * per] - x[i, -1]) if (i + 1) * per < 256 and t[w, (i + 1) * per] > 0 else -1
        print(f"  tile {i}: mfma {d[0]:6d} | " + " ".join(f"park {d[1+2*k]:5d} stream {d[2+2*k]:6d}" for k in range(4)) + f" | to next {nxt}")
