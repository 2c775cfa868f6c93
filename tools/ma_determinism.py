#!/usr/bin/env python
"""Run-to-run determinism of the memory attention at bench size (16 objects, Nk = 28736): the kernels have no atomics, so
two runs on the same inputs must agree bit for bit - a difference is a race.  python tools/ma_determinism.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from det_sam2_amd.build_sam import resolve_config
from det_sam2_amd.weights import synthetic_state_dict
from det_sam2_amd.hip_model import HipSam2
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = resolve_config("sam2.1_hiera_t")
sd = synthetic_state_dict(cfg, 0)
hm = HipSam2(cfg, sd, "cuda:0", max_batch=16)
hm.set_precision("bf16x3k")
g = torch.Generator().manual_seed(21)
B, NF, NP = 16, 7, 16
d = hm.device
curr = torch.randn(4096, 256, generator=g).to(d)
feats = [torch.randn(B, 64, 64, 64, generator=g).to(torch.bfloat16) for _ in range(NF)]
ptrs = [torch.randn(B, 256, generator=g) for _ in range(NP)]
mem_d, pos_d = hm.bank_assemble(B, [(f.flatten(2).transpose(1, 2).contiguous().to(d), r) for f, r in zip(feats, [6, 5, 4, 3, 2, 1, 0])],
                                [(p.to(d), q / 15.0) for p, q in zip(ptrs, range(NP))])
first = None
for r in range(reps):
    out = hm.memory_attention(B, curr, mem_d, pos_d, 4 * NP).clone()
    torch.cuda.synchronize()
    if first is None:
        first = out
        print("rep 0: checksum", float(out.double().sum()), "finite", bool(torch.isfinite(out).all()))
    else:
        diff = (out - first).abs()
        nbad = int((diff > 0).sum())
        print(f"rep {r}: {'identical' if nbad == 0 else 'DIFFERENT'}  elements differing {nbad}  max|d| {float(diff.max()):.3e}  rel {float(diff.norm() / first.norm()):.3e}", flush=True)
