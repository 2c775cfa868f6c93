#!/usr/bin/env python
"""Kernel time of the fused MLP (gemm_mlp256.hip) alone, on fixed random operands, under the default library and A/B builds (GPU box):
    python tools/mlp_time.py [NAME ...]     -> us per launch for the memory-attention FFN shape (65536 x 256 -> 2048 -> 256, ReLU, residual,
two fp16 terms) and the CXBlock shape (65536 x 1024, GELU, layer scale), HIP-event brackets of the library around the kernel launch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from det_sam2_amd.hip_model import HipOps
o = HipOps("cuda:0")
o.set_precision("bf16x3")
res = []
for rows, H, act, gam in ((65536, 2048, 1, False), (65536, 1024, 2, True)):
    g = torch.Generator().manual_seed(rows + H)
    X = torch.randn(rows, 256, generator=g).cuda()
    W1 = (torch.randn(H, 256, generator=g) * 0.06).cuda(); b1 = (torch.randn(H, generator=g) * 0.1).cuda()
    W2 = (torch.randn(256, H, generator=g) * 0.03).cuda(); b2 = (torch.randn(256, generator=g) * 0.1).cuda()
    R = torch.randn(rows, 256, generator=g).cuda()
    gm = torch.rand(256, generator=g).cuda() if gam else None
    for _ in range(3):
        o.op_mlp(X, W1, b1, W2, b2, gm, R, act)
    torch.cuda.synchronize()
    o.profile_enable(True, gemm_shapes=True)
    for t in o.profile_tags():
        o.profile_read(t)
    for _ in range(20):
        o.op_mlp(X, W1, b1, W2, b2, gm, R, act)
    torch.cuda.synchronize()
    best = None
    for t in o.profile_tags():
        ms, n = o.profile_read(t)
        if "mlp256" in t and n > 0 and "merge" not in t:
            best = (ms / n * 1e3, t)
    o.profile_enable(False)
    res.append("%%.1f us (%%s)" %% best if best else "no tag")
print("MLPTIME", " | ".join(res))
'''


def main():
    names = sys.argv[1:] or []
    for n in ["default"] + names:
        env = dict(os.environ, DS2_OP_MLP_F16X2="1")
        if n != "default":
            env["DS2_LIB"] = os.path.join(ROOT, "det-sam2_amd", "lib", f"ab_{n}.so")
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("MLPTIME")]
        print(f"{n:10s}", line[0] if line else r.stderr[-500:], flush=True)


if __name__ == "__main__":
    main()
