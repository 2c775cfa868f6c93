"""GPU parity of the primitive HIP ops (through the C-ABI) against fp64 PyTorch-CPU formulas."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from _util import record, rel_err


@pytest.fixture(scope="module", params=["fp32", "bf16x3"])
def ops(request):
    from det_sam2_amd.hip_model import HipOps
    o = HipOps("cuda:0")
    o.set_precision(request.param)
    o.tol = {"fp32": 2e-5, "bf16x3": 3e-4}[request.param]     # bf16x3: ~2^-16 relative error per product
    yield o
    o.set_precision("bf16x3")


@pytest.mark.parametrize("M,N,K,act,use_r,r_mod,use_g", [
    (128, 128, 64, 0, False, 0, False),
    (200, 96, 148, 0, True, 0, False),
    (1000, 432, 144, 2, False, 0, False),
    (513, 4, 256, 3, False, 0, False),
    (16, 1, 256, 0, False, 0, False),
    (300, 256, 576, 1, True, 100, True),
    (4096, 768, 256, 0, False, 0, False),
    (70000, 1152, 288, 2, False, 0, False),      # 256x256 LDS-DMA kernel (k_gemm_split_d256), ragged M and N tiles
    (66000, 432, 144, 2, False, 0, True),        # persistent 256x256 kernel (k_gemm_split_pp256): ragged N tile (432 of 512), 5 K tiles
    (40960, 576, 576, 0, True, 0, True),
    (9000, 256, 64, 0, True, 4500, False),       # K = 64 weight-stationary kernel (k_gemm_split_k64), ragged last row tile
    (4101, 128, 64, 1, False, 0, True),
])
def test_gemm(ops, M, N, K, act, use_r, r_mod, use_g):
    g = torch.Generator().manual_seed(M * 7 + N)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    R = torch.randn(r_mod if r_mod else M, N, generator=g) if use_r else None
    gam = torch.randn(N, generator=g) if use_g else None
    ref = A.double() @ W.double().T + b.double()
    ref = [lambda x: x, F.relu, F.gelu, torch.sigmoid][act](ref)
    if use_g:
        ref = ref * gam.double()
    if use_r:
        ref = ref + (R.double()[torch.arange(M) % r_mod] if r_mod else R.double())
    d = ops.device
    got = ops.op_gemm(A.to(d), W.to(d), b.to(d), act, None if gam is None else gam.to(d), None if R is None else R.to(d), r_mod)
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    record("gemm", prec=ops.get_precision(), M=M, N=N, K=K, act=act, err=e)
    assert e < ops.tol, e


@pytest.mark.parametrize("rows,H,act,use_r,use_g", [
    (128, 128, 1, False, False),         # one row block, one chunk pair
    (300, 256, 0, True, False),          # ragged last row block (300 = 2 x 128 + 44)
    (1000, 1024, 2, True, True),         # CXBlock shape: GELU, layer scale, residual
    (40000, 2048, 1, True, False),       # memory-attention FFN shape, more row blocks than CUs (persistent loop)
    (4096, 2048, 1, True, False),        # few rows: hidden dimension split over workgroups (4 chunks per part) + merge kernel
])
@pytest.mark.parametrize("f16x2", [False, True])
def test_fused_mlp(rows, H, act, use_r, use_g, f16x2, monkeypatch):
    """gemm_mlp256.hip (hidden activations in registers, W2 with the hidden index permuted inside 16-groups) against
    the fp64 formula; bf16x3 arithmetic.  f16x2: the two-term fp16 form (x and the hidden activations rounded to one fp16
    plane, weights two planes) the memory attention / memory encoder use in mode bf16x3k."""
    from det_sam2_amd.hip_model import HipOps
    monkeypatch.setenv("DS2_OP_MLP_F16X2", "1" if f16x2 else "0")
    o = HipOps("cuda:0")
    o.set_precision("bf16x3")
    g = torch.Generator().manual_seed(rows + H)
    X, W1, b1 = torch.randn(rows, 256, generator=g), torch.randn(H, 256, generator=g) / 16, torch.randn(H, generator=g)
    W2, b2 = torch.randn(256, H, generator=g) / math.sqrt(H), torch.randn(256, generator=g)
    R = torch.randn(rows, 256, generator=g) if use_r else None
    gam = torch.randn(256, generator=g) if use_g else None
    hid = X.double() @ W1.double().T + b1.double()
    hid = [lambda x: x, F.relu, F.gelu][act](hid)
    ref = hid @ W2.double().T + b2.double()
    if use_g:
        ref = ref * gam.double()
    if use_r:
        ref = ref + R.double()
    d = o.device
    got = o.op_mlp(X.to(d), W1.to(d), b1.to(d), W2.to(d), b2.to(d), None if gam is None else gam.to(d), None if R is None else R.to(d), act)
    again = o.op_mlp(X.to(d), W1.to(d), b1.to(d), W2.to(d), b2.to(d), None if gam is None else gam.to(d), None if R is None else R.to(d), act)
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    record("fused_mlp", rows=rows, H=H, act=act, f16x2=f16x2, err=e)
    assert e < (6e-4 if f16x2 else 3e-4), e       # (fp16 activations: 2^-12 relative per element)
    assert torch.equal(got, again)                 # run-to-run bit identity (DMA ring hazards show up here)


def test_fused_mlp_f16x2_saturates(monkeypatch):
    """The fp16 operand planes of the two-term form saturate: an activation beyond fp16's range (6.5e4) gives a finite result
    (the row it sits in is approximate), never inf / NaN, and the other rows are untouched."""
    from det_sam2_amd.hip_model import HipOps
    monkeypatch.setenv("DS2_OP_MLP_F16X2", "1")
    o = HipOps("cuda:0")
    o.set_precision("bf16x3")
    g = torch.Generator().manual_seed(3)
    X, W1, b1 = torch.randn(256, 256, generator=g), torch.randn(128, 256, generator=g) / 16, torch.randn(128, generator=g)
    W2, b2 = torch.randn(256, 128, generator=g) / 11, torch.randn(256, generator=g)
    Xbig = X.clone()
    Xbig[7, 5] = 3.0e6
    d = o.device
    ref = o.op_mlp(X.to(d), W1.to(d), b1.to(d), W2.to(d), b2.to(d), None, None, 1)
    got = o.op_mlp(Xbig.to(d), W1.to(d), b1.to(d), W2.to(d), b2.to(d), None, None, 1)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    keep = torch.ones(256, dtype=torch.bool)
    keep[7] = False
    assert torch.equal(got[keep.to(d)], ref[keep.to(d)])


@pytest.mark.parametrize("M,N,K,act,use_r,r_mod", [
    (128, 256, 256, 0, True, 0),       # token-side projection, 16 objects x 8 tokens, residual
    (128, 2048, 256, 1, False, 0),     # two-way transformer MLP, first layer (ReLU)
    (128, 256, 2048, 0, True, 0),      # ... second layer: K split over 16 waves
    (9, 128, 256, 0, False, 0),        # one object, ragged row group
    (24, 4, 256, 3, False, 0),         # fewer columns than a wave (sigmoid head)
    (40, 100, 64, 2, True, 8),         # ragged columns, GELU, broadcast residual
    (1, 256, 128, 0, False, 0),
])
def test_linear_small(ops, M, N, K, act, use_r, r_mod):
    """gemm_skinny.hip against the fp64 formula: exact fp32 arithmetic, so the bound is fp32 rounding; run-to-run and
    batch-composition bit identity (a row's result must not depend on the rows around it: sharded streams)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    gam = torch.randn(N, generator=g)
    R = torch.randn(r_mod if r_mod else M, N, generator=g) if use_r else None
    ref = A.double() @ W.double().T + b.double()
    ref = [lambda x: x, F.relu, F.gelu, torch.sigmoid][act](ref) * gam.double()
    if use_r:
        ref = ref + (R.double()[torch.arange(M) % r_mod] if r_mod else R.double())
    d = ops.device
    dev = lambda t: None if t is None else t.to(d)
    got = ops.op_linear_small(dev(A), dev(W), dev(b), act, dev(gam), dev(R), r_mod)
    again = ops.op_linear_small(dev(A), dev(W), dev(b), act, dev(gam), dev(R), r_mod)
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    record("linear_small", M=M, N=N, K=K, act=act, err=e)
    assert e < 2e-6, e
    assert torch.equal(got, again)
    if M > 1 and not use_r:     # the second half of the rows alone: same bits
        sub = ops.op_linear_small(dev(A[M // 2:].contiguous()), dev(W), dev(b), act, dev(gam))
        torch.cuda.synchronize()
        assert torch.equal(sub, got[M // 2:])


@pytest.mark.parametrize("rows,C,act", [(37, 96, 0), (1000, 256, 2), (5, 1152, 0), (64, 4, 2),
                                        (1003, 64, 2)])     # C = 64: k_layernorm_c64 (16 lanes per row), ragged last block
def test_layernorm(ops, rows, C, act):
    g = torch.Generator().manual_seed(rows)
    x, w, b = torch.randn(rows, C, generator=g) * 3 + 1, torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-6)
    if act == 2:
        ref = F.gelu(ref)
    got = ops.op_layernorm(x.to(ops.device), w.to(ops.device), b.to(ops.device), 1e-6, act)
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    record("layernorm", rows=rows, C=C, err=e)
    assert e < 1e-5, e


@pytest.mark.parametrize("B,H,D,DV,Lq,Lk", [
    (2, 1, 256, 256, 200, 300),
    (2, 1, 256, 64, 130, 1000),
    (3, 8, 16, 16, 9, 500),
    (3, 8, 16, 16, 9, 4096),    # few queries x many keys: split-key path (attention_fewq.hip)
    (2, 8, 16, 16, 16, 1100),   # ... ragged last key chunk, Lq at the path's maximum
    (2, 4, 32, 32, 7, 1024),
    (3, 8, 16, 16, 500, 9),
    (2, 8, 32, 32, 9, 9),
    (1, 2, 72, 72, 256, 256),
    (1, 2, 96, 96, 64, 64),
    (2, 4, 56, 56, 100, 196),
    (2, 8, 72, 72, 512, 1024),  # Hiera global attention: pre-split operands, attention_hg.hip (split modes)
    (1, 2, 96, 96, 256, 320),   # ... head dim 96 (tiny / small), odd tile count
    (1, 4, 56, 56, 256, 64),    # ... head dim 56 (base_plus), two key tiles
])
def test_attention_plain(ops, B, H, D, DV, Lq, Lk):
    g = torch.Generator().manual_seed(D * 31 + Lq)
    q, k, v = torch.randn(B, Lq, H * D, generator=g), torch.randn(B, Lk, H * D, generator=g), torch.randn(B, Lk, H * DV, generator=g)
    sc = 1.0 / math.sqrt(D)

    def heads(x, d):
        return x.double().reshape(B, -1, H, d).transpose(1, 2)

    ref = F.scaled_dot_product_attention(heads(q, D), heads(k, D), heads(v, DV), scale=sc).transpose(1, 2).reshape(B, Lq, H * DV)
    d = ops.device
    got = ops.op_attention(q.to(d), k.to(d), v.to(d), H, sc)
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    record("attention_plain", prec=ops.get_precision(), D=D, DV=DV, Lq=Lq, Lk=Lk, err=e)
    assert e < ops.tol, e


def _window_ref(q_nat, k_nat, v_nat, kb, vb, side_q, side_k, win_q, win_k, heads):
    """Reference windowed attention with zero-pad semantics (pad keys = bias rows)."""
    def part(x, side, ws, pad_row):
        C = x.shape[-1]
        x = x.reshape(side, side, C)
        p = (ws - side % ws) % ws
        if p:
            full = pad_row.reshape(1, 1, C).expand(side + p, side + p, C).clone()
            full[:side, :side] = x
            x = full
        sp = side + p
        x = x.reshape(sp // ws, ws, sp // ws, ws, C).permute(0, 2, 1, 3, 4).reshape(-1, ws * ws, C)
        return x, sp

    qw, spq = part(q_nat, side_q, win_q, torch.zeros(q_nat.shape[-1], dtype=q_nat.dtype))
    kw, _ = part(k_nat, side_k, win_k, kb)
    vw, _ = part(v_nat, side_k, win_k, vb)
    nW = qw.shape[0]
    D = q_nat.shape[-1] // heads

    def hd(x):
        return x.reshape(nW, -1, heads, D).transpose(1, 2)

    o = F.scaled_dot_product_attention(hd(qw), hd(kw), hd(vw)).transpose(1, 2).reshape(nW, win_q * win_q, -1)
    n = spq // win_q
    o = o.reshape(n, n, win_q, win_q, -1).permute(0, 2, 1, 3, 4).reshape(spq, spq, -1)
    return o[:side_q, :side_q].reshape(side_q * side_q, -1)


@pytest.mark.parametrize("side,win,heads,D,pool", [
    (64, 8, 1, 96, False), (64, 8, 2, 72, True), (64, 14, 4, 96, False), (64, 14, 2, 56, True), (32, 7, 2, 96, False),
    (64, 16, 8, 72, False),
    # hiera_l stage 3: 14 x 14 windows, head dim 72 - the whole-window kernel attention_win14.hip in the split modes
    (64, 14, 8, 72, False), (28, 14, 2, 72, False),
    # hiera_t / hiera_s stage 3: 14 x 14 windows, head dim 96 - the two-phase kernel (K planes, then V^T planes, in the same LDS)
    (28, 14, 3, 96, False),
    # small windows (16 / 64 keys): register-only kernel attention_smallwin.hip in bf16x3 mode
    (64, 4, 4, 72, False), (64, 4, 2, 72, True), (32, 8, 2, 56, False), (64, 8, 3, 96, True), (16, 4, 1, 96, False),
    (64, 4, 2, 56, True), (64, 4, 2, 56, False), (64, 8, 2, 56, True), (64, 8, 2, 96, False), (64, 4, 3, 96, True),
])
def test_attention_windowed(ops, side, win, heads, D, pool):
    g = torch.Generator().manual_seed(side + win)
    dim = heads * D
    qkv = torch.randn(side * side, 3 * dim, generator=g)
    bias = torch.randn(3 * dim, generator=g)
    k_nat, v_nat = qkv[:, dim:2 * dim], qkv[:, 2 * dim:]
    if pool:
        q_nat = F.max_pool2d(qkv[:, :dim].reshape(side, side, dim).permute(2, 0, 1)[None], 2, 2)[0].permute(1, 2, 0).reshape(-1, dim).contiguous()
        side_q, win_q = side // 2, win // 2
    else:
        q_nat, side_q, win_q = qkv[:, :dim], side, win
    ref = _window_ref(q_nat.double(), k_nat.double(), v_nat.double(), bias[dim:2 * dim].double(), bias[2 * dim:].double(),
                      side_q, side, win_q, win, heads)
    d = ops.device
    qkv_d, bias_d = qkv.to(d), bias.to(d)
    qd = q_nat.to(d).contiguous() if pool else qkv_d[:, :dim]
    nw = -(-side // win)
    got = ops.op_attention(qd, qkv_d[:, dim:2 * dim], qkv_d[:, 2 * dim:], heads, 1.0 / math.sqrt(D), win_q=win_q, win_k=win,
                           hq=side_q, wq=side_q, hk=side, wk=side, nwx=nw, k_pad=bias_d[dim:2 * dim], v_pad=bias_d[2 * dim:],
                           batch=nw * nw, lq=win_q * win_q, lk=win * win, dv=D)
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    record("attention_windowed", prec=ops.get_precision(), side=side, win=win, heads=heads, D=D, pool=pool, err=e)
    assert e < ops.tol, e
