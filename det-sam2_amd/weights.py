"""Checkpoint schema and the build-owned deterministic synthetic checkpoint generator.

``param_shapes(cfg)`` lists every ``state_dict`` entry the reference's strict
``load_state_dict`` expects (``sam2/build_sam.py:166-177``; key inventory in SURVEY.md
section 8b), so real SAM 2.1 checkpoints and the synthetic one go through the same door.

``synthetic_state_dict(cfg, seed)`` is a pure function of (parameter name, shape, seed):
no SAM 2.1 weights exist in this environment (no network), and torch's module-constructor
RNG order cannot be reproduced without the reference, so parity tests, goldens, ``smoke()``
and ``bench.py`` all use this generator (numpy Philox keyed by crc32(name)).
"""
from __future__ import annotations

import re
import zlib
from collections import OrderedDict

import numpy as np

from .config import ModelCfg, resolve_config


def _mlp(prefix, dims, out):
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        out[f"{prefix}.layers.{i}.weight"] = (b, a)
        out[f"{prefix}.layers.{i}.bias"] = (b,)


def _attn(prefix, emb, internal, kv_in, out):
    out[f"{prefix}.q_proj.weight"] = (internal, emb)
    out[f"{prefix}.q_proj.bias"] = (internal,)
    out[f"{prefix}.k_proj.weight"] = (internal, kv_in)
    out[f"{prefix}.k_proj.bias"] = (internal,)
    out[f"{prefix}.v_proj.weight"] = (internal, kv_in)
    out[f"{prefix}.v_proj.bias"] = (internal,)
    out[f"{prefix}.out_proj.weight"] = (emb, internal)
    out[f"{prefix}.out_proj.bias"] = (emb,)


def _ln(prefix, c, out):
    out[f"{prefix}.weight"] = (c,)
    out[f"{prefix}.bias"] = (c,)


def param_shapes(cfg) -> "OrderedDict[str, tuple]":
    cfg = resolve_config(cfg)
    o: "OrderedDict[str, tuple]" = OrderedDict()
    D, M = cfg.d_model, cfg.mem_dim
    # --- top-level SAM2Base parameters (sam2_base.py:136-176)
    o["maskmem_tpos_enc"] = (cfg.num_maskmem, 1, 1, M)
    o["no_mem_embed"] = (1, 1, D)
    o["no_mem_pos_enc"] = (1, 1, D)
    o["no_obj_ptr"] = (1, D)
    o["no_obj_embed_spatial"] = (1, M)
    # --- image encoder trunk (hieradet.py:172-267)
    t = cfg.trunk
    p = "image_encoder.trunk"
    o[f"{p}.pos_embed"] = (1, t.embed_dim, *t.window_pos_embed_bkg_spatial_size)
    o[f"{p}.pos_embed_window"] = (1, t.embed_dim, t.window_spec[0], t.window_spec[0])
    o[f"{p}.patch_embed.proj.weight"] = (t.embed_dim, 3, 7, 7)
    o[f"{p}.patch_embed.proj.bias"] = (t.embed_dim,)
    for i, b in enumerate(t.blocks()):
        q = f"{p}.blocks.{i}"
        _ln(f"{q}.norm1", b["dim"], o)
        o[f"{q}.attn.qkv.weight"] = (3 * b["dim_out"], b["dim"])
        o[f"{q}.attn.qkv.bias"] = (3 * b["dim_out"],)
        o[f"{q}.attn.proj.weight"] = (b["dim_out"], b["dim_out"])
        o[f"{q}.attn.proj.bias"] = (b["dim_out"],)
        _ln(f"{q}.norm2", b["dim_out"], o)
        _mlp(f"{q}.mlp", [b["dim_out"], 4 * b["dim_out"], b["dim_out"]], o)
        if b["dim"] != b["dim_out"]:
            o[f"{q}.proj.weight"] = (b["dim_out"], b["dim"])
            o[f"{q}.proj.bias"] = (b["dim_out"],)
    # --- FPN neck (image_encoder.py:75-90)
    for n, c in enumerate(t.channel_list):
        o[f"image_encoder.neck.convs.{n}.conv.weight"] = (D, c, 1, 1)
        o[f"image_encoder.neck.convs.{n}.conv.bias"] = (D,)
    # --- memory attention (memory_attention.py:17-118)
    for l in range(cfg.mem_attn_layers):
        q = f"memory_attention.layers.{l}"
        _attn(f"{q}.self_attn", D, D, D, o)
        _attn(f"{q}.cross_attn_image", D, D, M, o)
        o[f"{q}.linear1.weight"] = (cfg.mem_attn_ffn, D)
        o[f"{q}.linear1.bias"] = (cfg.mem_attn_ffn,)
        o[f"{q}.linear2.weight"] = (D, cfg.mem_attn_ffn)
        o[f"{q}.linear2.bias"] = (D,)
        for k in (1, 2, 3):
            _ln(f"{q}.norm{k}", D, o)
    _ln("memory_attention.norm", D, o)
    # --- memory encoder (memory_encoder.py:17-181)
    cin = 1
    for s in range(4):
        cout = cin * 4
        o[f"memory_encoder.mask_downsampler.encoder.{3 * s}.weight"] = (cout, cin, 3, 3)
        o[f"memory_encoder.mask_downsampler.encoder.{3 * s}.bias"] = (cout,)
        _ln(f"memory_encoder.mask_downsampler.encoder.{3 * s + 1}", cout, o)
        cin = cout
    o["memory_encoder.mask_downsampler.encoder.12.weight"] = (D, cin, 1, 1)
    o["memory_encoder.mask_downsampler.encoder.12.bias"] = (D,)
    o["memory_encoder.pix_feat_proj.weight"] = (D, D, 1, 1)
    o["memory_encoder.pix_feat_proj.bias"] = (D,)
    for l in range(2):
        q = f"memory_encoder.fuser.layers.{l}"
        o[f"{q}.gamma"] = (D,)
        o[f"{q}.dwconv.weight"] = (D, 1, 7, 7)
        o[f"{q}.dwconv.bias"] = (D,)
        _ln(f"{q}.norm", D, o)
        o[f"{q}.pwconv1.weight"] = (4 * D, D)
        o[f"{q}.pwconv1.bias"] = (4 * D,)
        o[f"{q}.pwconv2.weight"] = (D, 4 * D)
        o[f"{q}.pwconv2.bias"] = (D,)
    o["memory_encoder.out_proj.weight"] = (M, D, 1, 1)
    o["memory_encoder.out_proj.bias"] = (M,)
    # --- prompt encoder (prompt_encoder.py:17-62)
    o["sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = (2, D // 2)
    for n in range(4):
        o[f"sam_prompt_encoder.point_embeddings.{n}.weight"] = (1, D)
    o["sam_prompt_encoder.not_a_point_embed.weight"] = (1, D)
    o["sam_prompt_encoder.mask_downscaling.0.weight"] = (4, 1, 2, 2)
    o["sam_prompt_encoder.mask_downscaling.0.bias"] = (4,)
    _ln("sam_prompt_encoder.mask_downscaling.1", 4, o)
    o["sam_prompt_encoder.mask_downscaling.3.weight"] = (16, 4, 2, 2)
    o["sam_prompt_encoder.mask_downscaling.3.bias"] = (16,)
    _ln("sam_prompt_encoder.mask_downscaling.4", 16, o)
    o["sam_prompt_encoder.mask_downscaling.6.weight"] = (D, 16, 1, 1)
    o["sam_prompt_encoder.mask_downscaling.6.bias"] = (D,)
    o["sam_prompt_encoder.no_mask_embed.weight"] = (1, D)
    # --- mask decoder (mask_decoder.py:50-101, transformer.py:44-215)
    md = "sam_mask_decoder"
    for l in range(2):
        q = f"{md}.transformer.layers.{l}"
        _attn(f"{q}.self_attn", D, D, D, o)
        _ln(f"{q}.norm1", D, o)
        _attn(f"{q}.cross_attn_token_to_image", D, D // 2, D, o)
        _ln(f"{q}.norm2", D, o)
        _mlp(f"{q}.mlp", [D, 2048, D], o)
        _ln(f"{q}.norm3", D, o)
        _ln(f"{q}.norm4", D, o)
        _attn(f"{q}.cross_attn_image_to_token", D, D // 2, D, o)
    _attn(f"{md}.transformer.final_attn_token_to_image", D, D // 2, D, o)
    _ln(f"{md}.transformer.norm_final_attn", D, o)
    o[f"{md}.iou_token.weight"] = (1, D)
    o[f"{md}.mask_tokens.weight"] = (4, D)
    o[f"{md}.obj_score_token.weight"] = (1, D)
    o[f"{md}.output_upscaling.0.weight"] = (D, D // 4, 2, 2)
    o[f"{md}.output_upscaling.0.bias"] = (D // 4,)
    _ln(f"{md}.output_upscaling.1", D // 4, o)
    o[f"{md}.output_upscaling.3.weight"] = (D // 4, D // 8, 2, 2)
    o[f"{md}.output_upscaling.3.bias"] = (D // 8,)
    o[f"{md}.conv_s0.weight"] = (D // 8, D, 1, 1)
    o[f"{md}.conv_s0.bias"] = (D // 8,)
    o[f"{md}.conv_s1.weight"] = (D // 4, D, 1, 1)
    o[f"{md}.conv_s1.bias"] = (D // 4,)
    for n in range(4):
        _mlp(f"{md}.output_hypernetworks_mlps.{n}", [D, D, D, D // 8], o)
    _mlp(f"{md}.iou_prediction_head", [D, 256, 256, 4], o)
    _mlp(f"{md}.pred_obj_score_head", [D, D, D, 1], o)
    # --- pointers (sam2_base.py:236-252, 80)
    o["mask_downsample.weight"] = (1, 1, 4, 4)
    o["mask_downsample.bias"] = (1,)
    _mlp("obj_ptr_proj", [D, D, D, D], o)
    o["obj_ptr_tpos_proj.weight"] = (M, D)
    o["obj_ptr_tpos_proj.bias"] = (M,)
    return o


def _is_norm_name(name: str) -> bool:
    parts = name.split(".")
    leaf_owner = parts[-2] if len(parts) >= 2 else ""
    if leaf_owner.startswith("norm") or leaf_owner == "norm_final_attn":
        return True
    # LayerNorm2d layers inside nn.Sequential containers (numeric names)
    if "mask_downsampler.encoder." in name and leaf_owner.isdigit() and int(leaf_owner) % 3 == 1 and int(leaf_owner) < 12:
        return True
    if "mask_downscaling." in name and leaf_owner in ("1", "4"):
        return True
    if "output_upscaling.1." in name:
        return True
    return False


def _fan_in(name: str, shape) -> int:
    if name.endswith("output_upscaling.0.weight") or name.endswith("output_upscaling.3.weight"):
        return shape[0]  # ConvTranspose2d weight is [Cin, Cout, kh, kw]; stride==kernel -> Cin taps/output
    f = 1
    for s in shape[1:]:
        f *= s
    return max(f, 1)


def synthetic_tensor(name: str, shape, seed: int = 0) -> np.ndarray:
    """One parameter of the synthetic checkpoint (float32, C-contiguous)."""
    rng = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))
    n = rng.standard_normal(tuple(shape), dtype=np.float32)
    leaf = name.split(".")[-1]
    if _is_norm_name(name):
        out = 1.0 + 0.1 * n if leaf == "weight" else 0.05 * n
    elif leaf == "gamma":
        out = 0.2 * (1.0 + 0.1 * n)                      # CXBlock layer scale (trained >> 1e-6 init)
    elif name.endswith("positional_encoding_gaussian_matrix"):
        out = n                                           # PositionEmbeddingRandom scale=1.0
    elif name.endswith("pred_obj_score_head.layers.2.bias"):
        out = np.full(shape, 2.0, np.float32)             # keep objects "present" so masks are exercised
    elif leaf == "bias":
        out = 0.02 * n
    elif leaf == "weight" and len(shape) >= 2 and not (
        name.endswith("_embed.weight") or name.endswith("_token.weight") or name.endswith("_tokens.weight")
        or "point_embeddings" in name
    ):
        out = n / np.sqrt(np.float32(_fan_in(name, shape)))
    elif name in ("image_encoder.trunk.pos_embed", "image_encoder.trunk.pos_embed_window"):
        out = 0.1 * n
    else:                                                 # embeddings / tokens / tpos / no_mem / no_obj
        out = 0.3 * n
    return np.ascontiguousarray(out.astype(np.float32))


LOGIT_SCALE_KEYS = re.compile(r"sam_mask_decoder\.output_hypernetworks_mlps\.\d+\.layers\.2\.(weight|bias)$")


def synthetic_state_dict(cfg, seed: int = 0, logit_scale: float = 1.0):
    """OrderedDict[name -> torch.FloatTensor] (CPU) with exactly the keys of ``param_shapes``.

    ``logit_scale`` multiplies the last layer of the four hypernetwork MLPs (``mask_decoder.py:227-235``), i.e. every
    mask logit: the default weights give |logit| up to 14-17 (saturated sigmoids into the memory encoder); 1/30 gives the
    low-margin regime |logit| < 1 of the held-out goldens."""
    import torch

    cfg = resolve_config(cfg)
    out = OrderedDict()
    for k, s in param_shapes(cfg).items():
        a = synthetic_tensor(k, s, seed)
        if logit_scale != 1.0 and LOGIT_SCALE_KEYS.search(k):
            a = np.ascontiguousarray(a * np.float32(logit_scale))
        out[k] = torch.from_numpy(a)
    return out


def check_state_dict(cfg, sd) -> None:
    """Strict key/shape validation, mirroring ``_load_checkpoint`` (build_sam.py:166-177)."""
    want = param_shapes(cfg)
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    if missing or unexpected:
        raise RuntimeError(f"checkpoint mismatch: missing={missing[:8]} unexpected={unexpected[:8]}")
    for k, s in want.items():
        if tuple(sd[k].shape) != tuple(s):
            raise RuntimeError(f"checkpoint shape mismatch for {k}: {tuple(sd[k].shape)} != {tuple(s)}")
