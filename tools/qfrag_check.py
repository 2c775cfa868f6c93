#!/usr/bin/env python
"""The fused query kernel (gemm_qproj.hip) against the three kernels it replaces, at the level of the Q fragments (GPU box):
    python tools/qfrag_check.py [rows]
prints how many fp16 values differ and by how much (in units of the last place), per memory-attention layer, in mode bf16x3k."""
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det_sam2_amd.config import resolve_config  # noqa: E402
from det_sam2_amd.hip_model import HipSam2  # noqa: E402
from det_sam2_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    cfg = resolve_config("sam2.1_hiera_t")
    hm = HipSam2(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=4)
    hm.set_precision("bf16x3k")
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(rows, 256, generator=g) * 2 + 0.3).to(hm.device)
    ok = True
    for layer in range(4):
        a = hm.op_query_fragments(layer, x, False)
        b = hm.op_query_fragments(layer, x, True)
        torch.cuda.synchronize()
        fa, fb = a.view(torch.float16).float(), b.view(torch.float16).float()
        nd = int((a != b).sum())
        ulp = (a.int() - b.int()).abs()
        print(f"layer {layer}: {nd} of {a.numel()} fp16 values differ; max |diff| {float((fa - fb).abs().max()):.3e} (|q| max {float(fa.abs().max()):.2f}); "
              f"max distance {int(ulp.max())} in bit patterns; positions (first) {torch.nonzero(a != b)[:6].flatten().tolist()}")
        ok &= nd == 0
    print("QFRAG CHECK", "PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
