"""Host-side constant folding done once at model load (numpy, float32): tables that the
reference recomputes or caches at run time but that depend only on the checkpoint/config.
They are uploaded to the HIP library as '#'-named parameters (include/detsam2_hip.h).

All layouts are token-major ([H*W, C]).  Each function cites what it restates.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def _cubic_coeffs(t, A=F32(-0.75)):
    """Cubic convolution coefficients (PyTorch upsample_bicubic2d, A = -0.75)."""
    t = t.astype(F32)

    def cc1(x):  # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def cc2(x):  # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return np.stack([cc2(t + 1), cc1(t), cc1(1 - t), cc2(2 - t)], axis=-1).astype(F32)


def bicubic_resize(x, out_h, out_w):
    """F.interpolate(x[1,C,h,w], size=(out_h,out_w), mode='bicubic', align_corners=False)
    as used by Hiera._get_pos_embed (hieradet.py:276). Returns [C,out_h,out_w] float32."""
    x = np.asarray(x, F32)
    _, C, h, w = x.shape

    def axis_tables(n_in, n_out):
        scale = F32(n_in) / F32(n_out)
        src = scale * (np.arange(n_out, dtype=F32) + F32(0.5)) - F32(0.5)
        i0 = np.floor(src).astype(np.int64)
        coef = _cubic_coeffs(src - i0.astype(F32))
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)
        return idx, coef

    iy, cy = axis_tables(h, out_h)
    ix, cx = axis_tables(w, out_w)
    g = x[0][:, :, ix]                                    # [C,h,out_w,4]
    rows = (g * cx[None, None]).sum(-1, dtype=F32)        # interpolate along x
    g2 = rows[:, iy, :]                                   # [C,out_h,4,out_w]
    return (g2 * cy[None, :, :, None]).sum(2, dtype=F32)  # then along y


def hiera_pos_embed(pos_embed, pos_embed_window, side=256):
    """Hiera._get_pos_embed (hieradet.py:271-281) -> [side*side, C]."""
    pe = bicubic_resize(pos_embed, side, side)
    we = np.asarray(pos_embed_window, F32)[0]
    pe = pe + np.tile(we, (1, side // we.shape[1], side // we.shape[2]))
    return np.ascontiguousarray(pe.transpose(1, 2, 0).reshape(side * side, -1)).astype(F32)


def sine_pos_2d(num_pos_feats, h, w, temperature=10000.0):
    """PositionEmbeddingSine.forward, normalize=True (position_encoding.py:79-112) -> [h*w, num_pos_feats]."""
    npf = num_pos_feats // 2
    y = np.arange(1, h + 1, dtype=F32)[:, None].repeat(w, 1)
    x = np.arange(1, w + 1, dtype=F32)[None, :].repeat(h, 0)
    eps, scale = F32(1e-6), F32(2 * math.pi)
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = np.arange(npf, dtype=F32)
    dim_t = (F32(temperature) ** (F32(2) * np.floor(dim_t / 2) / F32(npf))).astype(F32)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = np.stack((np.sin(px[:, :, 0::2]), np.cos(px[:, :, 1::2])), axis=3).reshape(h, w, -1)
    py = np.stack((np.sin(py[:, :, 0::2]), np.cos(py[:, :, 1::2])), axis=3).reshape(h, w, -1)
    return np.ascontiguousarray(np.concatenate((py, px), axis=2).reshape(h * w, num_pos_feats)).astype(F32)


def rope_cis(dim=256, end_x=64, end_y=64, theta=10000.0):
    """compute_axial_cis (position_encoding.py:173-186) -> [end_x*end_y, dim/2, 2] (cos, sin)."""
    freqs = (F32(1.0) / (F32(theta) ** (np.arange(0, dim, 4)[: dim // 4].astype(F32) / F32(dim)))).astype(F32)
    t = np.arange(end_x * end_y, dtype=F32)
    tx, ty = np.mod(t, F32(end_x)), np.floor(t / F32(end_x))
    ang = np.concatenate([np.outer(tx, freqs), np.outer(ty, freqs)], axis=-1).astype(F32)
    return np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=-1)).astype(F32)


def dense_pe(gaussian, hw=64):
    """PromptEncoder.get_dense_pe (prompt_encoder.py:64-71; position_encoding.py:129-149) -> [hw*hw, 256]."""
    g = np.asarray(gaussian, F32)
    grid = np.ones((hw, hw), F32)
    y = (np.cumsum(grid, 0, dtype=F32) - F32(0.5)) / F32(hw)
    x = (np.cumsum(grid, 1, dtype=F32) - F32(0.5)) / F32(hw)
    c = np.stack([x, y], -1)
    c = (F32(2) * c - F32(1)) @ g
    c = (F32(2 * np.pi) * c).astype(F32)
    return np.ascontiguousarray(np.concatenate([np.sin(c), np.cos(c)], -1).reshape(hw * hw, -1)).astype(F32)


def ptr_dim_t(dim=256, temperature=10000.0):
    """Denominators of get_1d_sine_pe (sam2_utils.py:69-79) -> [dim/2]."""
    pe_dim = dim // 2
    d = np.arange(pe_dim, dtype=F32)
    return (F32(temperature) ** (F32(2) * np.floor(d / 2) / F32(pe_dim))).astype(F32)


def ingest_lut(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """load_video_frames storage chain (misc.py:328-342,358-359) as a [3,256] table of fp16 bit patterns:
    fp16(x/255.0) [float64 divide, stored into a float16 tensor], then `images -= mean; images /= std`
    on the float16 tensor with float32 mean/std (computed in fp32, rounded to fp16 after each op)."""
    v = np.arange(256, dtype=np.float64) / 255.0
    # torch converts double -> half through float (c10::Half(float)); replicate the double rounding
    h = v.astype(F32).astype(np.float16)
    out = np.empty((3, 256), np.uint16)
    for c in range(3):
        a = (h.astype(F32) - F32(mean[c])).astype(np.float16)
        b = (a.astype(F32) / F32(std[c])).astype(np.float16)
        out[c] = b.view(np.uint16)
    return out


def fold_out_v(w_out, b_out, w_v, b_v):
    """out_proj(v_proj(z)) = z (Wo Wv)^T + (Wo bv + bo): two Linear layers with nothing in between collapse into one
    (RoPEAttention / Attention.forward, sam/transformer.py:259-284: out_proj is applied to softmax(QK^T) V and the softmax
    rows sum to 1, so with V = M Wv^T + bv the product P V carries the bias through unchanged).  Folded in float64,
    rounded once to float32.  -> (W [256, kv_in], b [256])."""
    wo, wv = np.asarray(w_out, np.float64), np.asarray(w_v, np.float64)
    w = wo @ wv
    b = wo @ np.asarray(b_v, np.float64) + np.asarray(b_out, np.float64)
    return np.ascontiguousarray(w.astype(F32)), np.ascontiguousarray(b.astype(F32))


def model_constants(cfg, sd):
    """All '#'-named constants for ``cfg`` given the checkpoint ``sd`` (name -> numpy array)."""
    g = lambda k: np.asarray(sd[k], F32)  # noqa: E731
    hw = cfg.feat_hw
    folded = {}
    for l in range(cfg.mem_attn_layers):   # memory cross-attention: out_proj o v_proj as ONE 64 -> 256 projection
        p = f"memory_attention.layers.{l}.cross_attn_image."
        folded[f"#ma_cross_vo_w.{l}"], folded[f"#ma_cross_vo_b.{l}"] = fold_out_v(
            g(p + "out_proj.weight"), g(p + "out_proj.bias"), g(p + "v_proj.weight"), g(p + "v_proj.bias"))
        p = f"memory_attention.layers.{l}.self_attn."      # self-attention: the values are projected straight into the
        folded[f"#ma_self_vo_w.{l}"], folded[f"#ma_self_vo_b.{l}"] = fold_out_v(     # residual stream (no out_proj GEMM)
            g(p + "out_proj.weight"), g(p + "out_proj.bias"), g(p + "v_proj.weight"), g(p + "v_proj.bias"))
    return {
        **folded,
        "#pos_embed": hiera_pos_embed(g("image_encoder.trunk.pos_embed"), g("image_encoder.trunk.pos_embed_window"),
                                      cfg.image_size // 4),
        "#rope_cis": rope_cis(cfg.d_model, hw, hw, cfg.rope_theta),
        "#vision_pos": sine_pos_2d(cfg.d_model, hw, hw),
        "#maskmem_pos": sine_pos_2d(cfg.mem_dim, hw, hw),
        "#dense_pe": dense_pe(g("sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"), hw),
        "#ptr_dim_t": ptr_dim_t(cfg.d_model),
        "#ingest_lut": ingest_lut(),
    }
