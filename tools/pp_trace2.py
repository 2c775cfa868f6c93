#!/usr/bin/env python
"""Coarse per-tile timeline of the persistent GEMM kernel from a DS2_PP_TRACE=2 build: 17 stamps per tile, first 3 tiles of
workgroup 0:  DS2_LIB=.../ab_pptrace2.so DS2_GEMM_TILE=10 python tools/pp_trace2.py M N K [act]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from det_sam2_amd.hip_model import HipOps
from det_sam2_amd import _capi
ops = HipOps("cuda:0")
M, N, K = (int(x) for x in sys.argv[1:4])
act = int(sys.argv[4]) if len(sys.argv) > 4 else 2
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
for _ in range(3):
    ops.op_gemm(A, W, b, act)
torch.cuda.synchronize()
lib = ctypes.CDLL(_capi.LIB_PATH)
buf = np.zeros((8, 1024), dtype=np.uint64)
assert lib.ds2_debug_pp_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
names = ["start", "dma issued", "first landed", "loop done", "tail drained"] + [f"{e} tm{tm}" for tm in range(4) for e in ("parked", "stored", "freed")]
base = t[0, 0]
for w in (0, 2, 4, 6):
    n = int((t[w] > 0).sum()) // 17
    x = t[w, :n * 17].reshape(n, 17) - base
    print(f"wave {w} ({'loader' if not (w & 2) else 'storer'}):")
    for i in range(n):
        d = np.diff(x[i])
        print(f"  tile {i}: start @{x[i,0]:8d}  " + "  ".join(f"{names[k+1]}+{d[k]}" for k in range(16)) + (f"   | next tile starts +{t[w,(i+1)*17]-base-x[i,16]}" if i + 1 < n else ""))
