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


def build_sam2_video_predictor(config_file, ckpt_path=None, device="cuda", mode="eval", hydra_overrides_extra=[],
                               apply_postprocessing=True, **kwargs):
    cfg = resolve_config(config_file)
    if hydra_overrides_extra:
        raise NotImplementedError("hydra overrides are not interpreted; edit det_sam2_amd.config.ModelCfg instead")
    if not apply_postprocessing:
        raise NotImplementedError("apply_postprocessing=False (no dynamic multimask / binarised prompt masks) is not built")
    if mode != "eval":
        raise NotImplementedError("inference only (mode='eval')")
    dev = "cuda:0" if device in ("cuda", None) else str(device)
    # build_sam.py:134 appends ++model.fill_hole_area=8 when apply_postprocessing (hole filling of the low-res masks)
    return SAM2VideoPredictor(cfg, _load_state_dict(cfg, ckpt_path), device=dev, max_batch=kwargs.get("max_batch", 16),
                              fill_hole_area=kwargs.get("fill_hole_area", cfg.fill_hole_area))
