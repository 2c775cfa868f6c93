#!/usr/bin/env python
"""One GEMM shape, many launches (for rocprofv3 --kernel-trace --stats): python tools/op_bench1.py M N K [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from det_sam2_amd.hip_model import HipOps
ops = HipOps("cuda:0")
M, N, K = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
for _ in range(reps):
    ops.op_gemm(A, W, b)
torch.cuda.synchronize()
