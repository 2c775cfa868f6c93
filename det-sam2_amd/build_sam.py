"""build_sam2_video_predictor: same entry point as sam2/build_sam.py:111-146.

The reference composes a Hydra config and strict-loads ``torch.load(ckpt)["model"]``
(build_sam.py:166-177).  Here the config comes from ``det_sam2_amd.config`` (the reference's YAML
file names are accepted) and the checkpoint goes through the same strict key/shape check.  The five
overrides the reference appends for the video predictor (dynamic multimask via stability, binarised
prompt masks for the memory encoder, fill_hole_area) are part of ``ModelCfg``.
"""
from __future__ import annotations

import logging

from .config import resolve_config
from .sam2_video_predictor import SAM2VideoPredictor
from .weights import synthetic_state_dict


def _load_state_dict(cfg, ckpt_path):
    if ckpt_path is None:
        logging.warning("no checkpoint given: using the deterministic synthetic checkpoint (seed 0); "
                        "the reference would keep torch's random init here (build_sam.py:166-168)")
        return synthetic_state_dict(cfg, 0)
    if isinstance(ckpt_path, dict):
        return ckpt_path.get("model", ckpt_path)
    import torch

    return torch.load(ckpt_path, map_location="cpu")["model"]


# Hydra override keys (the reference's plugin mechanism, build_sam.py:121-141) that map onto ModelCfg fields.  Keys that
# select what this library IS (the predictor class, the five post-processing overrides at their reference values) are
# accepted when they ask for exactly that; anything that would need a different model raises.
_CFG_KEYS = {
    "model.fill_hole_area": ("fill_hole_area", int),
    "model.binarize_mask_from_pts_for_mem_enc": ("binarize_mask_from_pts_for_mem_enc", "bool"),
    "model.sam_mask_decoder_extra_args.dynamic_multimask_via_stability": ("dynamic_multimask_via_stability", "bool"),
    "model.sam_mask_decoder_extra_args.dynamic_multimask_stability_delta": ("dynamic_multimask_stability_delta", float),
    "model.sam_mask_decoder_extra_args.dynamic_multimask_stability_thresh": ("dynamic_multimask_stability_thresh", float),
    "model.max_cond_frames_in_attn": ("max_cond_frames_in_attn", int),
    "model.max_obj_ptrs_in_encoder": ("max_obj_ptrs_in_encoder", int),
    "model.sigmoid_scale_for_mem_enc": ("sigmoid_scale_for_mem_enc", float),
    "model.sigmoid_bias_for_mem_enc": ("sigmoid_bias_for_mem_enc", float),
    "model.multimask_min_pt_num": ("multimask_min_pt_num", int),
    "model.multimask_max_pt_num": ("multimask_max_pt_num", int),
}
_FIXED = {   # accepted only with the value the reference's video-predictor build uses
    "model._target_": "sam2.sam2_video_predictor.SAM2VideoPredictor",
    "model.image_size": 1024, "model.num_maskmem": 7,
}


def _parse_scalar(v):
    t = v.strip()
    if t.lower() in ("true", "false"):
        return t.lower() == "true"
    for cast in (int, float):
        try:
            return cast(t)
        except ValueError:
            pass
    return t.strip("'\"")


def apply_hydra_overrides(cfg, overrides):
    """``["++model.fill_hole_area=0", ...]`` -> ModelCfg with those fields replaced (dataclasses.replace)."""
    import dataclasses
    changes = {}
    for ov in overrides or []:
        key, sep, val = str(ov).lstrip("+~").partition("=")
        if not sep:
            raise ValueError(f"malformed hydra override {ov!r} (expected key=value)")
        key, val = key.strip(), _parse_scalar(val)
        if key in _CFG_KEYS:
            name, cast = _CFG_KEYS[key]
            changes[name] = bool(val) if cast == "bool" else cast(val)
        elif key in _FIXED:
            if val != _FIXED[key]:
                raise NotImplementedError(f"hydra override {ov!r}: only {key}={_FIXED[key]} is built")
        else:
            raise NotImplementedError(f"hydra override {ov!r} does not map onto det_sam2_amd.config.ModelCfg "
                                      f"(known keys: {sorted(list(_CFG_KEYS) + list(_FIXED))})")
    return dataclasses.replace(cfg, **changes) if changes else cfg


def resolve_build_cfg(config_file, hydra_overrides_extra=(), apply_postprocessing=True):
    """The ModelCfg `build_sam2_video_predictor` builds: config file, the caller's overrides, then - as build_sam.py:124-136
    APPENDS them after the caller's extras, and hydra lets the last override of a key win - the five postprocessing values."""
    import dataclasses
    cfg = resolve_config(config_file)
    if not apply_postprocessing:
        # without the five overrides of build_sam.py:126-135 the model keeps its constructor defaults: single-mask output is
        # token 0 (MaskDecoder.dynamic_multimask_via_stability = False, mask_decoder.py:37), prompted masks enter the memory
        # encoder through the sigmoid (SAM2Base.binarize_mask_from_pts_for_mem_enc = False), no hole filling
        cfg = dataclasses.replace(cfg, dynamic_multimask_via_stability=False, binarize_mask_from_pts_for_mem_enc=False, fill_hole_area=0)
    cfg = apply_hydra_overrides(cfg, list(hydra_overrides_extra))
    if apply_postprocessing:
        cfg = dataclasses.replace(cfg, dynamic_multimask_via_stability=True, dynamic_multimask_stability_delta=0.05,
                                  dynamic_multimask_stability_thresh=0.98, binarize_mask_from_pts_for_mem_enc=True, fill_hole_area=8)
    return cfg


def build_sam2_video_predictor(config_file, ckpt_path=None, device="cuda", mode="eval", hydra_overrides_extra=[],
                               apply_postprocessing=True, **kwargs):
    cfg = resolve_build_cfg(config_file, hydra_overrides_extra, apply_postprocessing)
    if mode != "eval":
        raise NotImplementedError("inference only (mode='eval')")
    dev = "cuda:0" if device in ("cuda", None) else str(device)
    # build_sam.py:134 appends ++model.fill_hole_area=8 when apply_postprocessing (hole filling of the low-res masks)
    return SAM2VideoPredictor(cfg, _load_state_dict(cfg, ckpt_path), device=dev, max_batch=kwargs.get("max_batch", 16),
                              fill_hole_area=kwargs.get("fill_hole_area", cfg.fill_hole_area))
