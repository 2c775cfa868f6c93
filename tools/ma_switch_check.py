#!/usr/bin/env python
"""Bit-identity of the memory attention under a runtime switch (GPU box):  python tools/ma_switch_check.py VAR=VALUE [B NF NP]
-> runs ds2_bank_memory_attention and ds2_memory_encoder on seeded inputs (sam2.1_hiera_t weights, mode bf16x3k) with and without the
variable set, in ONE process (the switches these checks are for are read per call), and compares the outputs bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det_sam2_amd.config import resolve_config  # noqa: E402
from det_sam2_amd.hip_model import HipSam2  # noqa: E402
from det_sam2_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    var, _, val = sys.argv[1].partition("=")
    B, NF, NP = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (16, 7, 16)
    cfg = resolve_config("sam2.1_hiera_t")
    hm = HipSam2(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=B)
    hm.set_precision("bf16x3k")
    g = torch.Generator().manual_seed(5)
    d = hm.device
    curr = torch.randn(4096, 256, generator=g).to(d)
    ents = [(torch.randn(B, 4096, 64, generator=g).to(torch.bfloat16).to(d), 6 - i) for i in range(NF)]
    ptrs = [(torch.randn(B, 256, generator=g).to(d), i / 15.0) for i in range(NP)]
    os.environ.pop(var, None)
    a = hm.bank_attention(B, curr, ents, ptrs).clone()
    os.environ[var] = val
    b = hm.bank_attention(B, curr, ents, ptrs).clone()
    os.environ.pop(var, None)
    c = hm.bank_attention(B, curr, ents, ptrs).clone()
    torch.cuda.synchronize()
    same = torch.equal(a, b) and torch.equal(a, c)
    print(f"memory attention  {var}={val} B={B} NF={NF} NP={NP}: max|diff| {float((a - b).abs().max()):.3e} (|out| max {float(a.abs().max()):.3f})",
          "SAME" if same else "DIFFERENT")
    f2 = torch.randn(4096, 256, generator=g).to(d)
    low = (torch.randn(B, 256, 256, generator=g) * 3).to(d)
    obj = torch.randn(B, generator=g).to(d)
    outs = []
    for setit in (False, True, False):
        if setit:
            os.environ[var] = val
        else:
            os.environ.pop(var, None)
        outs.append(hm.memory_encoder(B, f2, low, obj, False).clone())
    os.environ.pop(var, None)
    torch.cuda.synchronize()
    same2 = torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    print(f"memory encoder    {var}={val} B={B}: max|diff| {float((outs[0].float() - outs[1].float()).abs().max()):.3e}", "SAME" if same2 else "DIFFERENT")
    return 0 if same and same2 else 1


if __name__ == "__main__":
    sys.exit(main())
