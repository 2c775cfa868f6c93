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


def _e4m3(x):
    """OCP e4m3 (fn) of x, round to nearest even, saturating at +-448 - torch's CPU conversion saturates NaN-free inputs the same way"""
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)


def _mx_planes_ref(x, weight):
    """the documented MX plane pair of x (csrc/common.h): plane 1 = fp16(x) saturating; plane 2 = value byte | remainder byte << 8
    (activation) resp. remainder byte | value byte << 8 (weight), e4m3 of x * 2^-E and (x - fp16(x)) * 2^L"""
    EA, LA, EW, LW = -2, 14, -6, 18
    h = x.clamp(-65504.0, 65504.0).to(torch.float16)
    rem = x - h.float()
    v8 = _e4m3(x * 2.0 ** (-(EW if weight else EA))).view(torch.uint8).to(torch.int32)
    r8 = _e4m3(rem * 2.0 ** (LW if weight else LA)).view(torch.uint8).to(torch.int32)
    w = (r8 | (v8 << 8)) if weight else (v8 | (r8 << 8))
    return h.view(torch.int16), w


@pytest.mark.parametrize("weight", [False, True])
def test_mx_planes_are_the_documented_function_of_their_input(weight):
    """ds2_op_split_planes (the producers' conversion, ds2_mx_pair): every fp16 word and every e4m3 byte of the MX planes, bit for bit,
    against the definition in csrc/common.h - activations incl. values beyond the static e4m3 range (saturation, never a NaN byte),
    subnormal-range remainders, zeros; weights at the Hiera scale."""
    from det_sam2_amd.hip_model import HipOps
    ops = HipOps("cuda:0")
    g = torch.Generator().manual_seed(5 + weight)
    x = torch.randn(256, 576, generator=g) * (0.05 if weight else 1.0)
    x[::7, ::5] *= 50.0
    x[3, :8] = torch.tensor([0.0, -0.0, 1e-6, -3e-5, 65504.0, 1e6, -1e6, 447.9])
    p1, p2 = ops.op_split_planes(x.to(ops.device), 2 if weight else 1)
    torch.cuda.synchronize()
    r1, r2 = _mx_planes_ref(x, weight)
    assert torch.equal(p1.cpu()[:, :576], r1), int((p1.cpu()[:, :576] != r1).sum())
    got2 = p2.cpu()[:, :576].to(torch.int32) & 0xffff
    want2 = r2.to(torch.int32) & 0xffff
    assert torch.equal(got2, want2), int((got2 != want2).sum())
    assert not ((got2 & 0x7f) == 0x7f).any() and not (((got2 >> 8) & 0x7f) == 0x7f).any()        # no NaN byte, whatever the input


def test_mx_gelu_form_writes_mx_planes_of_the_fp64_result(monkeypatch):
    """The GELU epilogue of the MX form (k_gemm_x4gm_23_e2: v_cvt_pk_f16_f32, v_cvt_scalef32_pk_fp8_f32 x 2, v_perm_b32 under
    MODE.FP16_OVFL, written by the assembly drain) through ds2_op_gemm_planes in mode bf16x3k: the planes DECODE - fp16(plane 1) +
    e4m3(remainder byte) * 2^-14 - to gelu(A W^T + b) within the product's error, the value byte is e4m3(v * 4) of that value up to one
    code, and a run with outliers beyond the fp16 / e4m3 range saturates instead of producing inf / NaN codes."""
    from det_sam2_amd.hip_model import HipOps
    ops = HipOps("cuda:0")
    ops.set_precision("bf16x3k")
    d = ops.device
    M, N, K = 2048, 1152, 640
    g = torch.Generator().manual_seed(77)
    A = torch.randn(M, K, generator=g).to(d)
    W = (torch.randn(N, K, generator=g) * 0.05).to(d)
    b = torch.randn(N, generator=g).to(d)
    ops.profile_enable(True, gemm_shapes=True)
    for t in ops.profile_tags():
        ops.profile_read(t)
    hi, lo = ops.op_gemm_planes(A, W, b, 2)
    torch.cuda.synchronize()
    assert any(t.startswith("kern k_gemm_x4gm_23_e2") for t in ops.profile_tags())
    ops.profile_enable(False)
    exact = torch.nn.functional.gelu(A.double() @ W.double().T + b.double())
    h = hi.cpu().view(torch.float16).double()
    w2 = lo.cpu().to(torch.int32) & 0xffff
    rem = (w2 >> 8).to(torch.uint8).view(torch.float8_e4m3fn).double() * 2.0 ** -14
    val8 = (w2 & 0xff).to(torch.uint8).view(torch.float8_e4m3fn).double() * 2.0 ** -2
    dec = h + rem
    err = float((dec - exact.cpu()).norm() / exact.norm())
    assert err < 3e-5, err                                                  # (measured 1e-5: the product's error, the planes add 2^-15)
    # the value byte: e4m3 of the value scaled by 4 - within one e4m3 step (2^-3 relative) of it, or below the subnormal step
    ex = exact.cpu()
    assert bool(((val8 - ex).abs() <= 0.07 * ex.abs() + 2.0 ** -11).all())
    # saturation: a bias that drives outputs beyond fp16's and e4m3's ranges
    hi2, lo2 = ops.op_gemm_planes(A, W, b + 1e5, 2)
    torch.cuda.synchronize()
    h2 = hi2.cpu().view(torch.float16)
    assert torch.isfinite(h2.float()).all() and float(h2.float().max()) == 65504.0
    w22 = lo2.cpu().to(torch.int32) & 0xffff
    assert not ((w22 & 0x7f) == 0x7f).any() and not (((w22 >> 8) & 0x7f) == 0x7f).any()
    assert bool(((w22 & 0xff) == 0x7e).all())                               # every value byte saturated to +448
