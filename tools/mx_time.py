#!/usr/bin/env python
"""Kernel time of the MX form of the assembly GEMM (k_gemm_x4gm_23_e1 / e3) on the Hiera stage-3 / 4 shapes of sam2.1_hiera_l at a
16-frame batch, beside the three-term bf16 kernel of the same shapes (DS2_GEMM_MX=0) - the kernels alone (ds2_profile 'kern' brackets).
With an ablation build (tools/x4g_variant.py NAME flags...; DS2_LIB=det-sam2_amd/lib/ab_NAME.so) the MX column is that build's.

    python tools/mx_time.py [reps] [--nobf16]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from det_sam2_amd.hip_model import HipOps  # noqa: E402

SHAPES = [(65536, 1728, 576, 1, "s3 qkv"), (65536, 576, 576, 3, "s3 proj"), (65536, 576, 2304, 3, "s3 fc2"),
          (16384, 3456, 1152, 1, "s4 qkv"), (16384, 1152, 4608, 3, "s4 fc2")]


def timed(ops, fn, reps, prefix):
    fn()
    torch.cuda.synchronize()
    ops.profile_enable(True, gemm_shapes=True)
    for t in ops.profile_tags():
        ops.profile_read(t)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    best = None
    for t in ops.profile_tags():
        ms, n = ops.profile_read(t)
        if t.startswith("kern " + prefix) and n:
            best = (ms / n * 1e3, t.split()[1])
    ops.profile_enable(False)
    return best


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    ops = HipOps("cuda:0")
    ops.set_precision("bf16x3k")
    d = ops.device
    tot = [0.0, 0.0]
    for (M, N, K, form, nm) in SHAPES:
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(d)
        W = (torch.randn(N, K, generator=g) * 0.05).to(d)
        b = torch.randn(N, generator=g).to(d)
        R = torch.randn(M, N, generator=g).to(d) if form == 3 else None
        run = lambda: ops.op_gemm(A, W, b, 0, None, R, 0)  # noqa: E731
        os.environ["DS2_GEMM_MX"] = "1"
        t_mx = timed(ops, run, reps, "k_gemm_x4gm")
        line = f"{nm:8s} M={M} N={N} K={K} e{form}:  MX {t_mx[0]:7.1f} us ({2.0 * M * N * K / t_mx[0] * 1e-6:4.0f} TF)"
        tot[0] += t_mx[0]
        if "--nobf16" not in sys.argv:
            os.environ["DS2_GEMM_MX"] = "0"
            t_b = timed(ops, run, reps, "k_gemm_x4g_")
            line += f"   bf16x3 {t_b[0]:7.1f} us ({2.0 * M * N * K / t_b[0] * 1e-6:4.0f} TF)   x{t_b[0] / t_mx[0]:.2f}"
            tot[1] += t_b[0]
        print(line, flush=True)
    print(f"sum: MX {tot[0]:.1f} us" + (f"   bf16x3 {tot[1]:.1f} us" if tot[1] else ""))


if __name__ == "__main__":
    main()
