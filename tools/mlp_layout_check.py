#!/usr/bin/env python
"""Bit-identity of the fused MLP (gemm_mlp256.hip) between the default library and an A/B build (GPU box):
    python tools/mlp_layout_check.py NAME      -> sha256 of op_mlp results under lib/libdetsam2_hip.so and lib/ab_NAME.so
Shapes: the memory-attention FFN (65536 x 2048, ReLU, residual), the CXBlock MLP (65536 x 1024, GELU, layer scale + residual), a
few-row launch (hidden dimension split over workgroups), a ragged row count; three bf16 terms and two fp16 terms (DS2_OP_MLP_F16X2)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from det_sam2_amd.hip_model import HipOps
o = HipOps("cuda:0")
import os
for prec in ("f16x2" if os.environ.get("DS2_OP_MLP_F16X2") == "1" else "bf16x3",):
    o.set_precision("bf16x3")
    for rows, H, act, gam in ((65536, 2048, 1, False), (65536, 1024, 2, True), (4096, 2048, 1, False), (1000, 1024, 2, True)):
        g = torch.Generator().manual_seed(rows + H)
        X = torch.randn(rows, 256, generator=g).cuda()
        W1 = (torch.randn(H, 256, generator=g) * 0.06).cuda(); b1 = (torch.randn(H, generator=g) * 0.1).cuda()
        W2 = (torch.randn(256, H, generator=g) * 0.03).cuda(); b2 = (torch.randn(256, generator=g) * 0.1).cuda()
        R = torch.randn(rows, 256, generator=g).cuda()
        gm = torch.rand(256, generator=g).cuda() if gam else None
        y = o.op_mlp(X, W1, b1, W2, b2, gm, R, act)
        torch.cuda.synchronize()
        print(prec, rows, H, act, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16], float(y.abs().mean()))
'''


def main():
    name = sys.argv[1]
    outs = []
    for lib in (None, os.path.join(ROOT, "det-sam2_amd", "lib", f"ab_{name}.so")):
        lines = []
        for x2 in ("0", "1"):                      # three bf16 terms | two fp16 terms (the form the memory attention / memory encoder use)
            env = dict(os.environ, DS2_OP_MLP_F16X2=x2)
            if lib:
                env["DS2_LIB"] = lib
            r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True)
            got = [l for l in r.stdout.splitlines() if l.startswith(("bf16", "f16"))]
            if not got:
                print(r.stderr[-800:])
            lines += got
        outs.append(lines)
    ok = bool(outs[0]) and outs[0] == outs[1]
    for a, b in zip(*outs):
        print(a, "|", b.split()[4], "SAME" if a == b else "DIFFERENT")
    print("MLP LAYOUT CHECK", "PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
