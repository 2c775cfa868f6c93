#!/usr/bin/env python
"""Race / parity screen for a GEMM tile variant: the bf16x3 GEMM kernels share one per-element accumulation order, so two
variants must agree BIT FOR BIT.  Run on the GPU box:

    python tools/gemm_screen.py            # parent: runs the worker under DS2_GEMM_TILE=5 and =9, compares the outputs
    python tools/gemm_screen.py 5 9 --reps 10

The worker runs every shape `reps` times and also checks that all repetitions are identical (an LDS race shows up as a
result that comes and goes)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [  # M, N, K, act, residual, gamma
    (256, 256, 32, 0, 0, 0), (256, 256, 64, 0, 0, 0), (512, 512, 96, 0, 0, 0), (300, 200, 288, 2, 1, 1),
    (4096, 768, 256, 0, 0, 0), (40960, 576, 576, 0, 1, 1), (70000, 1152, 288, 2, 0, 0), (65536, 2304, 576, 2, 0, 0),
    (65536, 576, 2304, 0, 1, 0), (65536, 2048, 256, 1, 0, 0), (16384, 4608, 1152, 2, 0, 0), (65536, 1728, 576, 0, 0, 0),
    (1000, 1000, 1000 // 32 * 32 + 32, 0, 0, 0), (459776, 256, 64, 0, 0, 0), (65536, 256, 64, 2, 1, 1), (70001, 128, 64, 1, 0, 1),
]


def worker(tile, reps, out):
    import torch
    from det_sam2_amd.hip_model import HipOps
    ops = HipOps("cuda:0")
    d = ops.device
    res = {}
    for (M, N, K, act, use_r, use_g) in SHAPES:
        g = torch.Generator().manual_seed(M * 7 + N + K)
        A, W, b = torch.randn(M, K, generator=g).to(d), (torch.randn(N, K, generator=g) * 0.05).to(d), torch.randn(N, generator=g).to(d)
        R = torch.randn(M, N, generator=g).to(d) if use_r else None
        gam = torch.randn(N, generator=g).to(d) if use_g else None
        first = None
        bad = 0
        for _ in range(reps):
            got = ops.op_gemm(A, W, b, act, gam, R, 0)
            torch.cuda.synchronize()
            if first is None:
                first = got.clone()
            elif not torch.equal(first, got):
                bad += 1
        ref = (A.double() @ W.double().T + b.double())
        ref = [lambda x: x, torch.relu, torch.nn.functional.gelu, torch.sigmoid][act](ref)
        if use_g: ref = ref * gam.double()
        if use_r: ref = ref + R.double()
        err = float((first.double() - ref).norm() / ref.norm())
        print(f"tile {tile} M={M} N={N} K={K} act={act}: rel_err {err:.2e} unstable_reps {bad}/{reps - 1}", flush=True)
        res[(M, N, K)] = first.cpu()
    torch.save(res, out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    import torch
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 6
    tiles = [int(a) for a in args[:2]] or [5, 9]
    outs = []
    for t in tiles:
        out = f"/tmp/gemm_screen_{t}.pt"
        env = dict(os.environ)
        if t > 0:
            env["DS2_GEMM_TILE"] = str(t)      # tile 0 = the default dispatch (incl. the K = 64 kernel)
        subprocess.run([sys.executable, __file__, "--worker", str(t), str(reps), out], env=env, check=True)
        outs.append(torch.load(out))
    ok = True
    for k in outs[0]:
        same = torch.equal(outs[0][k], outs[1][k])
        md = float((outs[0][k] - outs[1][k]).abs().max())
        print(f"compare {k}: {'BIT-IDENTICAL' if same else 'DIFFERENT max|d|=%g' % md}")
        ok &= same
    print("SCREEN", "PASS" if ok else "FAIL")
    sys.exit(0 if ok else 1)
