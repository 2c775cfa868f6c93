"""GEMM ablation driver: run under rocprofv3 with DS2_GEMM_DBG=<mask> (1 no LDS stores, 2 no global loads,
4 no MFMA/ds_read, 8 no epilogue) and read the k_gemm_split times per shape from the kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from det_sam2_amd.hip_model import HipOps
ops = HipOps("cuda:0"); ops.set_precision("bf16x3"); d = ops.device
for (M, N, K) in [(16384, 2304, 576), (65536, 2048, 256), (65536, 256, 2048), (24576, 576, 576)]:
    A = torch.randn(M, K, device=d); W = torch.randn(N, K, device=d); b = torch.randn(N, device=d)
    for _ in range(4): ops.op_gemm(A, W, b)
torch.cuda.synchronize()
