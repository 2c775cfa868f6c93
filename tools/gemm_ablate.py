import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from det_sam2_amd.hip_model import HipOps
ops = HipOps("cuda:0"); ops.set_precision("bf16x3"); d = ops.device
for (M, N, K) in [(65536, 256, 2048), (4096, 2304, 576), (65536, 2048, 256)]:
    A = torch.randn(M, K, device=d); W = torch.randn(N, K, device=d); b = torch.randn(N, device=d)
    for _ in range(4): ops.op_gemm(A, W, b)
torch.cuda.synchronize()
