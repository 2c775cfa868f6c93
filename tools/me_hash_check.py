#!/usr/bin/env python
"""Bit-identity of the memory encoder between the default library and an A/B build (run on the GPU box):
python tools/me_hash_check.py NAME   -> sha256 of ds2_memory_encoder's output (sam2.1_hiera_t, 3 objects, both arithmetic modes)
under det-sam2_amd/lib/libdetsam2_hip.so and under det-sam2_amd/lib/ab_NAME.so."""
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
g = torch.Generator().manual_seed(9)
pix = torch.randn(4096, 256, generator=g).cuda()
low = (torch.randn(3, 256, 256, generator=g) * 3).cuda()
obj = torch.tensor([1.5, -0.5, 0.3]).cuda()
h = hashlib.sha256()
for prec in ("bf16x3k", "fp32"):
    hm.set_precision(prec)
    out = hm.memory_encoder(3, pix, low, obj, False)
    torch.cuda.synchronize()
    h.update(out.cpu().view(torch.int16).numpy().tobytes())
print(h.hexdigest())
''' % ROOT

if __name__ == "__main__":
    out = []
    for lib in (None, os.path.join(ROOT, "det-sam2_amd", "lib", f"ab_{sys.argv[1]}.so")):
        env = dict(os.environ)
        if lib:
            env["DS2_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        out.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-400:])
        print("default" if not lib else sys.argv[1], out[-1])
    print("bit-identical" if out[0] == out[1] and not out[0].startswith("FAILED") else "DIFFERENT")
