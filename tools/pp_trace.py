#!/usr/bin/env python
"""Segment timing of the persistent GEMM kernel from a DS2_PP_TRACE build (python tools/ab.py build pptrace -DDS2_PP_TRACE=1):
   DS2_LIB=.../ab_pptrace.so DS2_GEMM_TILE=10 python tools/pp_trace.py M N K
three stamps per step and wave: [end of the first segment | end of the second segment (barrier arrival) | barrier release];
waves 0-3: first = load, second = matrix; waves 4-7: first = matrix (previous phase), second = load."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from det_sam2_amd.hip_model import HipOps
from det_sam2_amd import _capi
ops = HipOps("cuda:0")
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
for _ in range(3):
    ops.op_gemm(A, W, b, 2)
torch.cuda.synchronize()
lib = ctypes.CDLL(_capi.LIB_PATH)
buf = np.zeros((8, 1024), dtype=np.uint64)
assert lib.ds2_debug_pp_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.astype(np.int64)
for w in range(8):
    n = int((t[w] > 0).sum())
    x = t[w, :n]
    if w >= 4:           # lagging group: drop the two stamps of its initial load-only step
        x = x[2:]
    n = len(x) // 3 * 3
    x = x[:n].reshape(-1, 3)
    first = x[1:, 0] - x[:-1, 2]; second = x[1:, 1] - x[1:, 0]; bar = x[1:, 2] - x[1:, 1]; per = x[1:, 2] - x[:-1, 2]
    s = slice(8, 64)
    print(f"wave {w}: first seg {first[s].mean():7.1f}  second seg {second[s].mean():7.1f}  barrier {bar[s].mean():7.1f}   step {per[s].mean():7.1f}   steps {len(per)}")
    if w in (0, 4):
        for i in range(8, 16):
            print(f"     step {i:3d} j={(i + 1) % 4}: first {first[i]:5d} second {second[i]:5d} barrier {bar[i]:5d}")
