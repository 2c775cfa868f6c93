#!/usr/bin/env python
"""Micro-benchmarks of the primitive ops on the GPU box (HIP events via torch): GEMM shapes of the workload."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from det_sam2_amd.hip_model import HipOps

ops = HipOps("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ops.set_precision(prec)
d = ops.device
shapes = [  # (M, N, K, what)
    (65536, 432, 144, "L s1 qkv"), (65536, 576, 144, "L s1 mlp0"), (65536, 144, 576, "L s1 mlp1"),
    (16384, 864, 288, "L s2 qkv"), (16384, 1152, 288, "L s2 mlp0"), (16384, 288, 1152, "L s2 mlp1"),
    (4096, 1728, 576, "L s3 qkv"), (4096, 576, 576, "L s3 proj"), (4096, 2304, 576, "L s3 mlp0"), (4096, 576, 2304, "L s3 mlp1"),
    (1024, 3456, 1152, "L s4 qkv"), (1024, 4608, 1152, "L s4 mlp0"), (1024, 1152, 4608, "L s4 mlp1"),
    (65536, 768, 256, "MA self qkv B16"), (65536, 256, 256, "MA proj B16"), (65536, 2048, 256, "MA ffn1"), (65536, 256, 2048, "MA ffn2"),
    (459776, 256, 64, "MA Kproj B16"), (65536, 256, 64, "MA vproj"), (65536, 32, 256, "conv_s0"), (65536, 256, 144, "neck0"),
    (262144, 128, 64, "dec up2 B16"), (65536, 1024, 256, "ME pw1"), (65536, 256, 1024, "ME pw2"), (16, 256, 256, "tiny mlp"),
]
for M, N, K, what in shapes:
    A = torch.randn(M, K, device=d); W = torch.randn(N, K, device=d); b = torch.randn(N, device=d)
    for _ in range(2): ops.op_gemm(A, W, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): ops.op_gemm(A, W, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{prec} gemm {what:18s} M={M:7d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF  blocks={-(-M//128)*-(-N//128)}")
