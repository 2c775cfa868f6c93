"""The assembly persistent GEMM (csrc/gemm_x4g.hip, epilogue of tile i under the main loop of tile i+1) against the other tile
kernels BIT FOR BIT - they share one per-element accumulation order and one epilogue arithmetic - and against the fp64 formula:
every epilogue form (bias | GELU + bf16 planes | bias + residual), both tile configurations (256 x 128, 128 x 192), one tile per
workgroup, several tiles per workgroup (drain bodies + plain bodies + tail), the minimum K (17 K tiles of 32 / 9 of 64) and longer K loops.
Shapes: the Linear layers of the Hiera blocks (sam2/modeling/backbones/hieradet.py:132-168)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # M, N, K, form
    (256, 384, 544, 1), (256, 384, 576, 2), (256, 384, 576, 3),            # one tile per workgroup, 17 / 18 K tiles
    (2048, 1152, 576, 1), (2048, 1152, 640, 2), (2048, 1152, 1152, 3),
    (65536, 1728, 576, 1),                                                 # qkv at a 16-frame batch: 18 / 24 tiles per workgroup
    (65536, 2304, 576, 2),                                                 # mlp.layers.0 (GELU, planes)
    (65536, 576, 2304, 3),                                                 # mlp.layers.1 (residual), 72 K tiles
    (16384, 1152, 1152, 3),
]


@pytest.fixture(scope="module")
def ops():
    from det_sam2_amd.hip_model import HipOps
    o = HipOps("cuda:0")
    o.set_precision("bf16x3")
    return o


def _run(ops, A, W, b, R, form):
    if form == 2:
        return torch.stack(ops.op_gemm_planes(A, W, b, 2))
    return ops.op_gemm(A, W, b, 0, None, R if form == 3 else None, 0)


@pytest.mark.parametrize("M,N,K,form", SHAPES)
def test_x4g_is_bit_identical_to_the_tile_kernels(ops, M, N, K, form, monkeypatch):
    d = ops.device
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + form)
    A = torch.randn(M, K, generator=g).to(d)
    W = (torch.randn(N, K, generator=g) * 0.05).to(d)
    b = torch.randn(N, generator=g).to(d)
    R = torch.randn(M, N, generator=g).to(d) if form == 3 else None
    monkeypatch.setenv("DS2_GEMM_X4G", "0")
    monkeypatch.delenv("DS2_GEMM_TILE", raising=False)
    ref = _run(ops, A, W, b, R, form)
    torch.cuda.synchronize()
    if form != 2 and M <= 4096:                                            # ... and the reference is itself right (fp64 formula)
        exact = A.double() @ W.double().T + b.double() + (R.double() if form == 3 else 0.0)
        assert float((ref.double() - exact).norm() / exact.norm()) < 3e-4
    ran = 0
    for tile, tm, tn in ((12, 256, 128), (13, 128, 192)):
        kt = 32 if tile == 12 else 64                                      # K tile of the configuration (gemm_x4g_body_*.inc)
        if M % tm or N % tn or K % kt or K // kt < (17 if tile == 12 else 9):
            continue
        monkeypatch.setenv("DS2_GEMM_TILE", str(tile))
        ops.profile_enable(True, gemm_shapes=True)
        for t in ops.profile_tags():
            ops.profile_read(t)
        got = _run(ops, A, W, b, R, form)
        again = _run(ops, A, W, b, R, form)
        torch.cuda.synchronize()
        used = [t for t in ops.profile_tags() if t.startswith("kern k_gemm_x4g")]
        ops.profile_enable(False)
        assert used, f"tile {tile}: the assembly kernel did not run (fell back)"
        assert torch.equal(got, ref), (tile, int((got != ref).sum()))
        assert torch.equal(again, got), "run-to-run difference (LDS race?)"
        ran += 1
    assert ran


def test_default_dispatch_takes_the_assembly_kernel_for_the_hiera_mlp(ops, monkeypatch):
    """the heuristic picks gemm_x4g for a chip-filling Hiera MLP shape (and DS2_GEMM_X4G=0 does not)"""
    d = ops.device
    A, W, b = torch.randn(16384, 576, device=d), torch.randn(2304, 576, device=d) * 0.05, torch.randn(2304, device=d)
    for env, want in (("1", True), ("0", False)):
        monkeypatch.setenv("DS2_GEMM_X4G", env)
        monkeypatch.delenv("DS2_GEMM_TILE", raising=False)
        ops.profile_enable(True, gemm_shapes=True)
        for t in ops.profile_tags():
            ops.profile_read(t)
        ops.op_gemm_planes(A, W, b, 2)
        torch.cuda.synchronize()
        used = any(t.startswith("kern k_gemm_x4g") for t in ops.profile_tags())
        ops.profile_enable(False)
        assert used == want


MX_SHAPES = [  # M, N, K, form (1: bias, 3: bias + residual)
    (128, 192, 576, 1), (256, 384, 576, 3), (2048, 1152, 640, 1), (2048, 1152, 1152, 3),
    (65536, 1728, 576, 1), (65536, 576, 2304, 3), (16384, 1152, 1152, 3), (4096, 576, 576, 3),
]


@pytest.mark.parametrize("outliers", [0.0, 20.0, 60.0])
@pytest.mark.parametrize("M,N,K,form", MX_SHAPES)
def test_mx_form_against_fp64(M, N, K, form, outliers, monkeypatch):
    """The two-MFMA-equivalent product of mode bf16x3k (gemm_x4g.hip "23m": fp16 hi.hi + both cross terms in ONE scaled fp8 MFMA per
    32 k's, static power-of-two scales - common.h "MX" planes) against the fp64 formula.  Error per product ~2^-15 (the e4m3 rounding
    of the 2^-11-sized remainders), i.e. ~2x the three-term bf16 product's; the same GEMM with DS2_GEMM_MX=0 is measured beside it.
    Operands: unit-variance activations, without / with outliers inside (x 20: |a| < ~100) / beyond (x 60: |a| up to ~300) the static
    e4m3 range of the value byte (|a| <= 112; remainder byte: |a| < ~56) - beyond it an element's cross terms saturate and its product
    degrades towards the plain fp16 one; weights of the Hiera scale (sigma 0.05)."""
    from det_sam2_amd.hip_model import HipOps
    ops = HipOps("cuda:0")
    ops.set_precision("bf16x3k")
    d = ops.device
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + form)
    A = torch.randn(M, K, generator=g)
    if outliers:
        A[::97, ::53] *= outliers
    A = A.to(d)
    W = (torch.randn(N, K, generator=g) * 0.05).to(d)
    b = torch.randn(N, generator=g).to(d)
    R = torch.randn(M, N, generator=g).to(d) if form == 3 else None
    rows = slice(0, min(M, 4096))
    exact = A[rows].double() @ W.double().T + b.double() + (R[rows].double() if form == 3 else 0.0)
    errs = {}
    for mx in ("1", "0"):
        monkeypatch.setenv("DS2_GEMM_MX", mx)
        ops.profile_enable(True, gemm_shapes=True)
        for t in ops.profile_tags():
            ops.profile_read(t)
        got = _run(ops, A, W, b, R, form)
        again = _run(ops, A, W, b, R, form)
        torch.cuda.synchronize()
        used = any(t.startswith("kern k_gemm_x4gm") for t in ops.profile_tags())
        ops.profile_enable(False)
        assert used == (mx == "1"), (mx, ops.profile_tags())
        assert torch.equal(got, again), "run-to-run difference (LDS race?)"
        assert torch.isfinite(got).all()
        errs[mx] = float((got[rows].double() - exact).norm() / exact.norm())
    from _util import record
    record("gemm_mx", M=M, N=N, K=K, form=form, outliers=outliers, rel_err_mx=errs["1"], rel_err_bf16x3=errs["0"])
    assert errs["1"] < (1.2e-5 if outliers <= 20.0 else 5e-5) and errs["0"] < 1e-5, errs
