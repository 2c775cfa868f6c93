#!/usr/bin/env python
"""Bit-identity of k_layernorm_c64 with k_layernorm_vec<1,false> (run on the GPU box; needs det-sam2_amd/lib/ab_noc64.so built
with -DDS2_LN_C64=0):  python tools/ln_c64_check.py   -> prints the sha256 of the result of both libraries."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from det_sam2_amd.hip_model import HipOps
o = HipOps("cuda:0")
g = torch.Generator().manual_seed(7)
x = (torch.randn(262144 + 3, 64, generator=g) * 3 + 1).cuda()
w, b = torch.randn(64, generator=g).cuda(), torch.randn(64, generator=g).cuda()
y = o.op_layernorm(x, w, b, 1e-6, 2)
torch.cuda.synchronize()
print(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())
''' % ROOT

if __name__ == "__main__":
    out = []
    for lib in (None, os.path.join(ROOT, "det-sam2_amd", "lib", "ab_noc64.so")):
        env = dict(os.environ)
        if lib:
            env["DS2_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        out.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:])
        print("default" if not lib else "noc64  ", out[-1])
    print("bit-identical" if out[0] == out[1] and not out[0].startswith("FAILED") else "DIFFERENT")
