#!/usr/bin/env python
"""Where does every output element of the assembly GEMM come from?  acc[m, n] = m + 1 (A[:, 0] = m + 1, W[:, 0] = 1), bias[n] = 1024 n:
C[m, n] = m + 1 + 1024 n decodes to its source (m, n); the output is pre-filled with -1 (unwritten elements show)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from det_sam2_amd import _capi
from det_sam2_amd.hip_model import HipOps, _p

ops = HipOps("cuda:0")
d = ops.device
M, N, K = 256, 384, 576
A = torch.zeros(M, K, device=d); A[:, 0] = torch.arange(1, M + 1, device=d).float()
W = torch.zeros(N, K, device=d); W[:, 0] = 1.0
b = (torch.arange(N, device=d) * 1024).float()
for tile in (12, 13):
    os.environ["DS2_GEMM_TILE"] = str(tile)
    out = torch.full((M, N), -1.0, device=d)
    _capi.check(ops.lib.ds2_op_gemm(M, N, K, _p(A), K, _p(W), K, _p(b), _p(out), N, 0, None, None, 0, 0, ops._stream()), "gemm")
    torch.cuda.synchronize()
    o = out.cpu().long()
    exp = (torch.arange(1, M + 1)[:, None] + 1024 * torch.arange(N)[None, :])
    print(f"tile {tile}: wrong {(o != exp).sum().item()} of {M * N}, unwritten {(o == -1).sum().item()}")
    sm, sn = (o % 1024) - 1, o // 1024
    for r in list(range(0, 12)) + [32, 33, 36, 128, 129]:
        print(f"  row {r:3d}: " + " ".join("   ok  " if o[r, c] == exp[r, c] else ("  ---  " if o[r, c] == -1 else f"{sm[r, c]:3d},{sn[r, c]:<3d}") for c in list(range(0, 10)) + [32, 33, 64, 65, 128]))
