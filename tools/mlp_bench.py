#!/usr/bin/env python
"""Micro-benchmark of the fused MLP kernel (gemm_mlp256.hip) through ds2_op_mlp: the memory-attention FFN shape and the
CXBlock shape at 16 objects.  DS2_LIB=<ab build> selects a variant (tools/ab.py build NAME -DDS2_MLP_ABL=n)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from det_sam2_amd.hip_model import HipOps

o = HipOps("cuda:0")
o.set_precision("bf16x3")
d = o.device
for rows, H, act in ((65536, 2048, 1), (65536, 1024, 2)):
    g = torch.Generator().manual_seed(1)
    X, W1, b1 = torch.randn(rows, 256, generator=g).to(d), (torch.randn(H, 256, generator=g) / 16).to(d), torch.randn(H, generator=g).to(d)
    W2, b2, R = (torch.randn(256, H, generator=g) / math.sqrt(H)).to(d), torch.randn(256, generator=g).to(d), torch.randn(rows, 256, generator=g).to(d)
    for _ in range(3):
        o.op_mlp(X, W1, b1, W2, b2, None, R, act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        o.op_mlp(X, W1, b1, W2, b2, None, R, act)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n          # (includes the operand-split pre-pass of X: ~25 us)
    print(f"{os.environ.get('DS2_LIB', 'default').split('/')[-1]:14s} mlp rows={rows} H={H} act={act}: {ms * 1e3:8.1f} us  {2.0 * rows * 256 * 2 * H / ms / 1e9:7.1f} TFLOP/s")
