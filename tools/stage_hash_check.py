#!/usr/bin/env python
"""Bit-identity of the tracking-chain stages between the default library and an A/B build (run on the GPU box):
python tools/stage_hash_check.py NAME  -> sha256 per stage (sam2.1_hiera_t, 3 objects, bf16x3k and fp32) under
det-sam2_amd/lib/libdetsam2_hip.so and under det-sam2_amd/lib/ab_NAME.so: the SAM heads (tracked-frame form, conditioning-frame
form with a box prompt, mask-prompt form), the memory attention (shared layer-0 tokens, 2 memory frames + 3 pointer tokens) and the
memory encoder."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from det_sam2_amd.config import resolve_config
from det_sam2_amd.hip_model import HipSam2
from det_sam2_amd.weights import synthetic_state_dict
cfg = resolve_config("sam2.1_hiera_t")
hm = HipSam2(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=4)
g = torch.Generator().manual_seed(11)
B = 3
pix = torch.randn(B, 4096, 256, generator=g).cuda()
pix1 = torch.randn(4096, 256, generator=g).cuda()
f0 = torch.randn(65536, 32, generator=g).cuda()
f1 = torch.randn(16384, 64, generator=g).cuda()
low = (torch.randn(B, 256, 256, generator=g) * 3).cuda()
obj = torch.tensor([1.5, -0.5, 0.3]).cuda()
box = torch.tensor([[[100., 120.], [400., 500.]], [[10., 20.], [900., 800.]], [[300., 300.], [600., 640.]]]).cuda()
lab = torch.tensor([[2, 3]] * B, dtype=torch.int32).cuda()
nk = 2 * 4096 + 12
mem = torch.randn(B, nk, 64, generator=g).to(torch.bfloat16).float().cuda()
mpos = torch.randn(B, nk, 64, generator=g).cuda()
def h(ts):
    d = hashlib.sha256()
    for t in ts:
        d.update(t.detach().float().cpu().numpy().tobytes())
    return d.hexdigest()[:16]
for prec in ("bf16x3k", "fp32"):
    hm.set_precision(prec)
    out = {}
    out["heads_tracked"] = h(hm.sam_heads(B, pix, f0, f1, None, None, multimask=True))
    out["heads_cond_box"] = h(hm.sam_heads(B, pix1, f0, f1, box, lab, multimask=False, pix_bcast=True, add_no_mem_embed=True))
    out["heads_mask"] = h(hm.sam_heads(B, pix1, f0, f1, None, None, multimask=False, pix_bcast=True, mask_inputs=low))
    out["memory_attention"] = h([hm.memory_attention(B, pix1, mem, mpos, 12)])
    out["memory_encoder"] = h([hm.memory_encoder(B, pix1, low, obj, False)])
    torch.cuda.synchronize()
    for k, v in out.items():
        print("HASH", prec, k, v)
''' % ROOT

if __name__ == "__main__":
    res = []
    for lib in (None, os.path.join(ROOT, "det-sam2_amd", "lib", f"ab_{sys.argv[1]}.so")):
        env = dict(os.environ)
        if lib:
            env["DS2_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
        if not lines:
            print("FAILED", r.stderr[-600:])
        res.append(lines)
    for a, b in zip(*res):
        print(a[5:], "|", b.split()[-1], "same" if a == b else "DIFFERENT")
