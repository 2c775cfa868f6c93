"""Is the persistent 256x256 GEMM power-limited?  The same [65536, 2304, 576] product on DS2_GEMM_NCU=256 / 128 workgroups
(one per CU); run under rocprofv3 --kernel-trace --stats and compare k_gemm_split_pp256's average duration."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from det_sam2_amd.hip_model import HipOps

ops = HipOps("cuda:0")
ops.set_precision("bf16x3")
d = ops.device
M, N, K = 65536, 2304, 576
A = torch.randn(M, K, device=d); W = torch.randn(N, K, device=d) / 24; b = torch.randn(N, device=d)
for _ in range(60):
    ops.op_gemm(A, W, b)
torch.cuda.synchronize()
