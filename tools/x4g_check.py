#!/usr/bin/env python
"""Parity + timing of the assembly GEMM (gemm_x4g.hip) against the other tile kernels, shape by shape (GPU box):

    python tools/x4g_check.py small          # a few one- and two-tile shapes, every epilogue form, both configurations
    python tools/x4g_check.py big [reps]     # the Hiera stage-3 / stage-4 shapes of sam2.1_hiera_l at a 16-frame batch + timings

Every comparison is BIT FOR BIT (the tile kernels share one per-element accumulation order and one epilogue arithmetic)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det_sam2_amd.hip_model import HipOps  # noqa: E402

SMALL = [  # M, N, K, form
    (256, 384, 576, 1), (256, 384, 576, 2), (256, 384, 576, 3), (512, 768, 544, 1), (512, 768, 640, 2), (512, 768, 1152, 3),
    (2048, 1152, 576, 1), (2048, 1152, 576, 2), (2048, 1152, 576, 3), (16384, 1152, 1152, 2),
]
BIG = [(65536, 1728, 576, 1), (65536, 2304, 576, 2), (65536, 576, 576, 3), (65536, 576, 2304, 3), (16384, 4608, 1152, 2),
       (16384, 1152, 4608, 3), (16384, 3456, 1152, 1), (16384, 1152, 1152, 3)]


def run(ops, A, W, b, R, form):
    if form == 1:
        return ops.op_gemm(A, W, b, 0, None, None, 0)
    if form == 3:
        return ops.op_gemm(A, W, b, 0, None, R, 0)
    hi, lo = ops.op_gemm_planes(A, W, b, 2)
    return torch.stack([hi, lo])


def timed(ops, fn, reps):
    """average duration of the GEMM KERNEL alone (library HIP-event brackets 'kern <kernel> M N K'), us, and its name"""
    fn()
    torch.cuda.synchronize()
    ops.profile_enable(True, gemm_shapes=True)
    for t in ops.profile_tags():
        ops.profile_read(t)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    best = (0.0, "?")
    for t in ops.profile_tags():
        ms, n = ops.profile_read(t)
        if t.startswith("kern ") and n > 0 and "split_rows" not in t:
            best = (ms / n * 1e3, t.split()[1])
    ops.profile_enable(False)
    return best


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 5
    nocheck = "--nocheck" in sys.argv          # timing of an ablation build (tools/x4g_variant.py): results are wrong by construction
    ops = HipOps("cuda:0")
    d = ops.device
    ok = True
    shapes = SMALL if which == "small" else BIG
    if "--only" in sys.argv:
        shapes = [shapes[int(i)] for i in sys.argv[sys.argv.index("--only") + 1].split(",")]
    for (M, N, K, form) in shapes:
        g = torch.Generator().manual_seed(M + 3 * N + 7 * K + form)
        A = torch.randn(M, K, generator=g).to(d)
        W = (torch.randn(N, K, generator=g) * 0.05).to(d)
        b = torch.randn(N, generator=g).to(d)
        R = torch.randn(M, N, generator=g).to(d) if form == 3 else None
        os.environ["DS2_GEMM_X4G"] = "0"
        os.environ.pop("DS2_GEMM_TILE", None)
        ref = run(ops, A, W, b, R, form)
        torch.cuda.synchronize()
        line = f"M={M} N={N} K={K} form e{form}:"
        t_ref, k_ref = timed(ops, lambda: run(ops, A, W, b, R, form), reps) if which != "small" else (0.0, "")
        for tile, nm in ((12, "256x128"), (13, "128x192")):
            if N % (128 if tile == 12 else 192) or M % (256 if tile == 12 else 128):
                continue
            os.environ["DS2_GEMM_TILE"] = str(tile)
            got = run(ops, A, W, b, R, form)
            torch.cuda.synchronize()
            same = nocheck or torch.equal(got, ref)
            bad = 0
            for _ in range(0 if nocheck else 2):          # run-to-run identity (an LDS race shows up as a result that comes and goes)
                again = run(ops, A, W, b, R, form)
                torch.cuda.synchronize()
                bad += int(not torch.equal(again, got))
            if not same:
                df = (got.float() - ref.float()) if form != 2 else (got != ref).float()
                nbad = int((df != 0).sum())
                rows = torch.nonzero((df != 0).reshape(-1, df.shape[-1]).any(1))[:8].flatten().tolist()
                cols = torch.nonzero((df != 0).reshape(-1, df.shape[-1]).any(0))[:8].flatten().tolist()
                line += f"  [{nm}: DIFFERENT {nbad} elements, rows {rows} cols {cols}]"
                g2, r2 = got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])
                for (ri, ci) in ((0, 0), (0, 1), (1, 0), (4, 0), (33, 33), (130, 70), (200, 150)):
                    if ri < g2.shape[0] and ci < g2.shape[1]:
                        line += f"\n      [{ri},{ci}] got {g2[ri, ci].item()} ref {r2[ri, ci].item()}"
                bm = (df != 0).reshape(-1, df.shape[-1])[:256, :384]
                line += "\n      bad fraction per 32x32 block (first 256 x 384): " + " | ".join(
                    " ".join(f"{bm[i:i + 32, j:j + 32].float().mean():.2f}" for j in range(0, bm.shape[1], 32)) for i in range(0, bm.shape[0], 32))
            else:
                line += f"  [{nm}: bit-identical"
                if which != "small":
                    t, k = timed(ops, lambda: run(ops, A, W, b, R, form), reps)
                    line += f" {k} {t:.1f} us ({2.0 * M * N * K / t * 1e-6:.0f} TF) vs {k_ref} {t_ref:.1f} us ({2.0 * M * N * K / t_ref * 1e-6:.0f} TF)"
                line += "]"
            if bad:
                line += f" UNSTABLE {bad}/2"
            ok &= same and not bad
        print(line, flush=True)
    print("X4G CHECK", "PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
