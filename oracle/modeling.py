"""ORACLE (test infrastructure, not product): fp32 PyTorch-CPU restatement of the neural
modules of the Det-SAM2 hot path.

Written functionally over a flat ``state_dict`` (``sd``) instead of nn.Modules.  Each
function cites the reference ``file:line`` it restates.  Pinned against the reference
itself: ``oracle/make_goldens.py`` imports the reference in the build container (through
``oracle/_ref_shims.py``) and commits its outputs under ``tests/golden``; ``tests/test_oracle_golden.py``
compares this file with them (all four configs).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg import this.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

NO_OBJ_SCORE = -1024.0  # sam2_base.py:21


# ----------------------------------------------------------------------------- helpers
def linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def layer_norm(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def layer_norm_2d(sd, p, x, eps=1e-6):
    """sam2_utils.py:150-162 (channel-first LayerNorm, biased variance)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + ".weight"][:, None, None] * x + sd[p + ".bias"][:, None, None]


def mlp(sd, p, x, n, act=F.relu, sigmoid_out=False):
    """sam2_utils.py:124-147."""
    for i in range(n):
        x = linear(sd, f"{p}.layers.{i}", x)
        if i < n - 1:
            x = act(x)
    return torch.sigmoid(x) if sigmoid_out else x


def sine_pos_2d(num_pos_feats, h, w, temperature=10000.0):
    """PositionEmbeddingSine.forward, normalize=True, scale=2*pi (position_encoding.py:79-112).
    Returns [num_pos_feats, h, w]."""
    npf = num_pos_feats // 2
    y = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).repeat(1, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).repeat(h, 1)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)


def sine_pe_1d(pos, dim, temperature=10000.0):
    """get_1d_sine_pe (sam2_utils.py:69-79)."""
    pe_dim = dim // 2
    dim_t = torch.arange(pe_dim, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / pe_dim)
    e = pos.unsqueeze(-1) / dim_t
    return torch.cat([e.sin(), e.cos()], dim=-1)


# ----------------------------------------------------------------------------- A5: Hiera
def _window_partition(x, ws):
    """backbones/utils.py:16-41 (zero padding to a multiple of ws)."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C), (Hp, Wp)


def _window_unpartition(w, ws, pad_hw, hw):
    """backbones/utils.py:44-66."""
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def _pool2(x):
    """do_pool with MaxPool2d(2,2) on a channels-last tensor (hieradet.py:25-36)."""
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def hiera_block(sd, p, blk, x):
    """MultiScaleBlock.forward + MultiScaleAttention.forward (hieradet.py:132-168, 57-82)."""
    shortcut = x
    x = layer_norm(sd, p + ".norm1", x, 1e-6)
    if blk["dim"] != blk["dim_out"]:
        shortcut = linear(sd, p + ".proj", x)
        if blk["q_stride"]:
            shortcut = _pool2(shortcut)
    ws = blk["window"]
    H, W = x.shape[1:3]
    if ws > 0:
        x, pad_hw = _window_partition(x, ws)
    Bw, h, w, _ = x.shape
    nh = blk["heads"]
    qkv = linear(sd, p + ".attn.qkv", x).reshape(Bw, h * w, 3, nh, -1)
    q, k, v = torch.unbind(qkv, 2)
    if blk["q_stride"]:
        q = _pool2(q.reshape(Bw, h, w, -1))
        h, w = q.shape[1:3]
        q = q.reshape(Bw, h * w, nh, -1)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    o = o.transpose(1, 2).reshape(Bw, h, w, -1)
    x = linear(sd, p + ".attn.proj", o)
    if blk["q_stride"]:
        ws = ws // blk["q_stride"]
        H, W = shortcut.shape[1:3]
        pad_hw = (H + (ws - H % ws) % ws, W + (ws - W % ws) % ws)
    if blk["window"] > 0:
        x = _window_unpartition(x, ws, pad_hw, (H, W))
    x = shortcut + x
    return x + mlp(sd, p + ".mlp", layer_norm(sd, p + ".norm2", x, 1e-6), 2, F.gelu)


def hiera_pos_embed(sd, p, hw):
    """Hiera._get_pos_embed (hieradet.py:271-281): bicubic background + tiled window embed."""
    we = sd[p + ".pos_embed_window"]
    pe = F.interpolate(sd[p + ".pos_embed"], size=hw, mode="bicubic")
    pe = pe + we.tile([a // b for a, b in zip(pe.shape, we.shape)])
    return pe.permute(0, 2, 3, 1)


def hiera_trunk(sd, cfg, img):
    """Hiera.forward (hieradet.py:283-299). img [1,3,S,S] -> list of 4 NCHW stage outputs."""
    p = "image_encoder.trunk"
    x = F.conv2d(img, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(sd, p, x.shape[1:3])
    outs = []
    ends = cfg.trunk.stage_ends
    for i, blk in enumerate(cfg.trunk.blocks()):
        x = hiera_block(sd, f"{p}.blocks.{i}", blk, x)
        if i in ends:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


# ----------------------------------------------------------------------------- A4: neck / forward_image
def fpn_neck(sd, cfg, xs):
    """FpnNeck.forward (image_encoder.py:101-134), nearest top-down on fpn_top_down_levels."""
    n = len(xs) - 1
    out, pos, prev = [None] * len(xs), [None] * len(xs), None
    for i in range(n, -1, -1):
        q = f"image_encoder.neck.convs.{n - i}.conv"
        lat = F.conv2d(xs[i], sd[q + ".weight"], sd[q + ".bias"])
        if i in cfg.fpn_top_down_levels and prev is not None:
            prev = lat + F.interpolate(prev.float(), scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
        pos[i] = sine_pos_2d(cfg.d_model, prev.shape[-2], prev.shape[-1])[None].repeat(prev.shape[0], 1, 1, 1)
    return out, pos


def forward_image(sd, cfg, img):
    """SAM2Base.forward_image (sam2_base.py:450-461) incl. ImageEncoder.forward scalp
    (image_encoder.py:30-43).  Returns (backbone_fpn[3], vision_pos_enc[3]) NCHW."""
    feats, pos = fpn_neck(sd, cfg, hiera_trunk(sd, cfg, img))
    if cfg.scalp > 0:
        feats, pos = feats[: -cfg.scalp], pos[: -cfg.scalp]
    md = "sam_mask_decoder"
    feats = list(feats)
    feats[0] = F.conv2d(feats[0], sd[md + ".conv_s0.weight"], sd[md + ".conv_s0.bias"])
    feats[1] = F.conv2d(feats[1], sd[md + ".conv_s1.weight"], sd[md + ".conv_s1.bias"])
    return feats, list(pos)


# ----------------------------------------------------------------------------- A12: memory attention
def axial_cis(dim, end_x, end_y, theta=10000.0):
    """compute_axial_cis (position_encoding.py:173-186)."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
    t = torch.arange(end_x * end_y, dtype=torch.float32)
    tx, ty = (t % end_x).float(), torch.div(t, end_x, rounding_mode="floor").float()
    fx, fy = torch.outer(tx, freqs), torch.outer(ty, freqs)
    return torch.cat([torch.polar(torch.ones_like(fx), fx), torch.polar(torch.ones_like(fy), fy)], dim=-1)


def apply_rope(xq, xk, cis, repeat_k):
    """apply_rotary_enc (position_encoding.py:196-220). x*: [B, heads, L, D]."""
    q_ = torch.view_as_complex(xq.float().reshape(*xq.shape[:-1], -1, 2))
    c = cis.view(1, 1, *cis.shape)
    q_out = torch.view_as_real(q_ * c).flatten(3)
    if xk.shape[-2] == 0:
        return q_out, xk
    k_ = torch.view_as_complex(xk.float().reshape(*xk.shape[:-1], -1, 2))
    if repeat_k:
        r = k_.shape[-2] // q_.shape[-2]
        c = c.unsqueeze(2).expand(-1, -1, r, -1, -1).flatten(2, 3)
    return q_out, torch.view_as_real(k_ * c).flatten(3)


def rope_attention(sd, p, q, k, v, cis, num_k_exclude_rope=0, repeat_k=False):
    """RoPEAttention.forward, num_heads=1 (transformer.py:312-363)."""
    q, k, v = linear(sd, p + ".q_proj", q), linear(sd, p + ".k_proj", k), linear(sd, p + ".v_proj", v)
    q, k, v = q.unsqueeze(1), k.unsqueeze(1).clone(), v.unsqueeze(1)
    nk = k.size(-2) - num_k_exclude_rope
    q, k[:, :, :nk] = apply_rope(q, k[:, :, :nk], cis, repeat_k)
    o = F.scaled_dot_product_attention(q, k, v)
    return linear(sd, p + ".out_proj", o.squeeze(1))


def memory_attention(sd, cfg, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens=0):
    """MemoryAttention.forward + MemoryAttentionLayer.forward (memory_attention.py:119-176, 59-99).
    curr/curr_pos [HW,B,256]; memory/memory_pos [Nk,B,64] -> [HW,B,256]."""
    out = (curr + 0.1 * curr_pos).transpose(0, 1)
    mem, mem_pos = memory.transpose(0, 1), memory_pos.transpose(0, 1)
    hw = int(math.isqrt(out.shape[1]))
    cis = axial_cis(cfg.d_model, hw, hw, cfg.rope_theta)
    for l in range(cfg.mem_attn_layers):
        p = f"memory_attention.layers.{l}"
        t2 = layer_norm(sd, p + ".norm1", out)
        out = out + rope_attention(sd, p + ".self_attn", t2, t2, t2, cis)
        t2 = layer_norm(sd, p + ".norm2", out)
        out = out + rope_attention(sd, p + ".cross_attn_image", t2, mem + mem_pos, mem, cis,
                                   num_k_exclude_rope=num_obj_ptr_tokens, repeat_k=True)
        t2 = layer_norm(sd, p + ".norm3", out)
        out = out + linear(sd, p + ".linear2", F.relu(linear(sd, p + ".linear1", t2)))
    return layer_norm(sd, "memory_attention.norm", out).transpose(0, 1)


# ----------------------------------------------------------------------------- A7: prompt encoder
def _pe_encoding(sd, coords01):
    """PositionEmbeddingRandom._pe_encoding (position_encoding.py:129-136)."""
    g = sd["sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = (2 * coords01 - 1) @ g
    c = 2 * math.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd, cfg):
    """PromptEncoder.get_dense_pe (prompt_encoder.py:64-71; position_encoding.py:138-149). [1,256,h,w]"""
    h = w = cfg.feat_hw
    grid = torch.ones((h, w), dtype=torch.float32)
    y = (grid.cumsum(0) - 0.5) / h
    x = (grid.cumsum(1) - 0.5) / w
    return _pe_encoding(sd, torch.stack([x, y], dim=-1)).permute(2, 0, 1).unsqueeze(0)


def prompt_encoder(sd, cfg, coords, labels, mask_prompt=None):
    """PromptEncoder.forward with points only (boxes=None => pad=True) (prompt_encoder.py:73-171)."""
    pe = "sam_prompt_encoder"
    B = coords.shape[0]
    pts = coords + 0.5
    pts = torch.cat([pts, torch.zeros((B, 1, 2))], dim=1)
    lab = torch.cat([labels, -torch.ones((B, 1), dtype=labels.dtype)], dim=1)
    c = pts.clone()
    c[:, :, 0] = c[:, :, 0] / cfg.image_size
    c[:, :, 1] = c[:, :, 1] / cfg.image_size
    emb = _pe_encoding(sd, c.to(torch.float))
    emb[lab == -1] = 0.0
    emb[lab == -1] += sd[pe + ".not_a_point_embed.weight"]
    for n in range(4):
        emb[lab == n] += sd[f"{pe}.point_embeddings.{n}.weight"]
    if mask_prompt is not None:
        d = F.conv2d(mask_prompt, sd[pe + ".mask_downscaling.0.weight"], sd[pe + ".mask_downscaling.0.bias"], stride=2)
        d = F.gelu(layer_norm_2d(sd, pe + ".mask_downscaling.1", d))
        d = F.conv2d(d, sd[pe + ".mask_downscaling.3.weight"], sd[pe + ".mask_downscaling.3.bias"], stride=2)
        d = F.gelu(layer_norm_2d(sd, pe + ".mask_downscaling.4", d))
        dense = F.conv2d(d, sd[pe + ".mask_downscaling.6.weight"], sd[pe + ".mask_downscaling.6.bias"])
    else:
        dense = sd[pe + ".no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, cfg.feat_hw, cfg.feat_hw)
    return emb, dense


# ----------------------------------------------------------------------------- A8: mask decoder
def _attention(sd, p, q, k, v, heads):
    """Attention.forward (transformer.py:239-284)."""
    q, k, v = linear(sd, p + ".q_proj", q), linear(sd, p + ".k_proj", k), linear(sd, p + ".v_proj", v)

    def sep(x):
        b, n, c = x.shape
        return x.reshape(b, n, heads, c // heads).transpose(1, 2)

    o = F.scaled_dot_product_attention(sep(q), sep(k), sep(v))
    b, h, n, c = o.shape
    return linear(sd, p + ".out_proj", o.transpose(1, 2).reshape(b, n, h * c))


def two_way_transformer(sd, src, pos_src, tokens):
    """TwoWayTransformer.forward + TwoWayAttentionBlock.forward (transformer.py:91-131, 182-215)."""
    p = "sam_mask_decoder.transformer"
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos_src.flatten(2).permute(0, 2, 1)
    queries, query_pe = tokens, tokens
    for l in range(2):
        q_ = f"{p}.layers.{l}"
        if l == 0:
            queries = _attention(sd, q_ + ".self_attn", queries, queries, queries, 8)
        else:
            qq = queries + query_pe
            queries = queries + _attention(sd, q_ + ".self_attn", qq, qq, queries, 8)
        queries = layer_norm(sd, q_ + ".norm1", queries)
        qq, kk = queries + query_pe, keys + key_pe
        queries = layer_norm(sd, q_ + ".norm2", queries + _attention(sd, q_ + ".cross_attn_token_to_image", qq, kk, keys, 8))
        queries = layer_norm(sd, q_ + ".norm3", queries + mlp(sd, q_ + ".mlp", queries, 2))
        qq, kk = queries + query_pe, keys + key_pe
        keys = layer_norm(sd, q_ + ".norm4", keys + _attention(sd, q_ + ".cross_attn_image_to_token", kk, qq, queries, 8))
    qq, kk = queries + query_pe, keys + key_pe
    queries = queries + _attention(sd, p + ".final_attn_token_to_image", qq, kk, keys, 8)
    return layer_norm(sd, p + ".norm_final_attn", queries), keys


def mask_decoder(sd, cfg, image_embeddings, image_pe, sparse, dense, multimask_output, high_res_features):
    """MaskDecoder.forward / predict_masks / dynamic multimask (mask_decoder.py:105-296),
    repeat_image=False, pred_obj_scores(+mlp), use_high_res_features, use_multimask_token_for_obj_ptr."""
    md = "sam_mask_decoder"
    B = sparse.size(0)
    out_tok = torch.cat([sd[md + ".obj_score_token.weight"], sd[md + ".iou_token.weight"], sd[md + ".mask_tokens.weight"]], 0)
    tokens = torch.cat((out_tok.unsqueeze(0).expand(B, -1, -1), sparse), dim=1)
    src = image_embeddings + dense
    pos_src = torch.repeat_interleave(image_pe, B, dim=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(sd, src, pos_src, tokens)
    iou_tok, mask_toks = hs[:, 1, :], hs[:, 2:6, :]
    src = src.transpose(1, 2).view(b, c, h, w)
    feat_s0, feat_s1 = high_res_features
    up = F.conv_transpose2d(src, sd[md + ".output_upscaling.0.weight"], sd[md + ".output_upscaling.0.bias"], stride=2)
    up = F.gelu(layer_norm_2d(sd, md + ".output_upscaling.1", up + feat_s1))
    up = F.conv_transpose2d(up, sd[md + ".output_upscaling.3.weight"], sd[md + ".output_upscaling.3.bias"], stride=2)
    up = F.gelu(up + feat_s0)
    hyper = torch.stack([mlp(sd, f"{md}.output_hypernetworks_mlps.{i}", mask_toks[:, i, :], 3) for i in range(4)], dim=1)
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
    iou = mlp(sd, md + ".iou_prediction_head", iou_tok, 3, sigmoid_out=True)
    obj_logits = mlp(sd, md + ".pred_obj_score_head", hs[:, 0, :], 3)
    if multimask_output:
        masks, iou = masks[:, 1:], iou[:, 1:]
        sam_tokens = mask_toks[:, 1:]
    elif not getattr(cfg, "dynamic_multimask_via_stability", True):   # mask_decoder.py:147-148
        masks, iou = masks[:, 0:1], iou[:, 0:1]
        sam_tokens = mask_toks[:, 0:1]
    else:
        # _dynamic_multimask_via_stability (mask_decoder.py:261-296)
        mm, mi = masks[:, 1:], iou[:, 1:]
        best = torch.argmax(mi, dim=-1)
        bi = torch.arange(B)
        best_m, best_i = mm[bi, best].unsqueeze(1), mi[bi, best].unsqueeze(1)
        sm, si = masks[:, 0:1], iou[:, 0:1]
        fl = sm.flatten(-2)
        d = cfg.dynamic_multimask_stability_delta
        ai, au = (fl > d).sum(-1).float(), (fl > -d).sum(-1).float()
        stab = torch.where(au > 0, ai / au, 1.0)
        ok = stab >= cfg.dynamic_multimask_stability_thresh
        masks = torch.where(ok[..., None, None].expand_as(sm), sm, best_m)
        iou = torch.where(ok.expand_as(si), si, best_i)
        sam_tokens = mask_toks[:, 0:1]
    return masks, iou, sam_tokens, obj_logits


# ----------------------------------------------------------------------------- A13: memory encoder
def memory_encoder(sd, cfg, pix_feat, masks):
    """MemoryEncoder.forward with skip_mask_sigmoid=True (memory_encoder.py:158-181),
    MaskDownSampler (17-58), Fuser/CXBlock (62-135). Returns (features [B,64,h,w], pos [B,64,h,w])."""
    me = "memory_encoder"
    x = masks
    for s in range(4):
        q = f"{me}.mask_downsampler.encoder.{3 * s}"
        x = F.conv2d(x, sd[q + ".weight"], sd[q + ".bias"], stride=2, padding=1)
        x = F.gelu(layer_norm_2d(sd, f"{me}.mask_downsampler.encoder.{3 * s + 1}", x))
    x = F.conv2d(x, sd[me + ".mask_downsampler.encoder.12.weight"], sd[me + ".mask_downsampler.encoder.12.bias"])
    x = F.conv2d(pix_feat, sd[me + ".pix_feat_proj.weight"], sd[me + ".pix_feat_proj.bias"]) + x
    for l in range(2):
        q = f"{me}.fuser.layers.{l}"
        inp = x
        x = F.conv2d(x, sd[q + ".dwconv.weight"], sd[q + ".dwconv.bias"], padding=3, groups=x.shape[1])
        x = layer_norm_2d(sd, q + ".norm", x).permute(0, 2, 3, 1)
        x = linear(sd, q + ".pwconv2", F.gelu(linear(sd, q + ".pwconv1", x)))
        x = (sd[q + ".gamma"] * x).permute(0, 3, 1, 2)
        x = inp + x
    x = F.conv2d(x, sd[me + ".out_proj.weight"], sd[me + ".out_proj.bias"])
    pos = sine_pos_2d(cfg.mem_dim, x.shape[-2], x.shape[-1])[None].repeat(x.shape[0], 1, 1, 1)
    return x, pos
