"""hydra_overrides_extra (the reference's plugin mechanism, sam2/build_sam.py:121-141) mapped onto ModelCfg - CPU only."""
import pytest

from det_sam2_amd.build_sam import apply_hydra_overrides
from det_sam2_amd.config import resolve_config


def test_reference_video_predictor_overrides_are_accepted():
    cfg = resolve_config("configs/sam2.1/sam2.1_hiera_l.yaml")
    ref = ["++model._target_=sam2.sam2_video_predictor.SAM2VideoPredictor",                      # build_sam.py:121-123
           "++model.sam_mask_decoder_extra_args.dynamic_multimask_via_stability=true",
           "++model.sam_mask_decoder_extra_args.dynamic_multimask_stability_delta=0.05",
           "++model.sam_mask_decoder_extra_args.dynamic_multimask_stability_thresh=0.98",
           "++model.binarize_mask_from_pts_for_mem_enc=true", "++model.fill_hole_area=8"]      # :126-135
    assert apply_hydra_overrides(cfg, ref) == cfg


def test_overrides_change_model_cfg():
    cfg = resolve_config("sam2.1_hiera_t")
    got = apply_hydra_overrides(cfg, ["++model.fill_hole_area=0", "model.max_cond_frames_in_attn=4",
                                      "++model.sam_mask_decoder_extra_args.dynamic_multimask_stability_thresh=0.9"])
    assert (got.fill_hole_area, got.max_cond_frames_in_attn, got.dynamic_multimask_stability_thresh) == (0, 4, 0.9)
    assert got.trunk == cfg.trunk and cfg.fill_hole_area == 8


def test_unsupported_overrides_fail_loudly():
    cfg = resolve_config("sam2.1_hiera_t")
    with pytest.raises(NotImplementedError, match="does not map"):
        apply_hydra_overrides(cfg, ["++model.memory_attention.num_layers=2"])
    with pytest.raises(NotImplementedError, match="only model.image_size=1024"):
        apply_hydra_overrides(cfg, ["++model.image_size=512"])
    with pytest.raises(ValueError):
        apply_hydra_overrides(cfg, ["model.fill_hole_area"])


def test_decoder_stability_override_maps_onto_cfg():
    cfg = resolve_config("sam2.1_hiera_t")
    got = apply_hydra_overrides(cfg, ["++model.sam_mask_decoder_extra_args.dynamic_multimask_via_stability=false"])
    assert cfg.dynamic_multimask_via_stability and not got.dynamic_multimask_via_stability


def test_postprocessing_overrides_win_over_the_callers_extras():
    """build_sam.py:124-136 appends the five postprocessing overrides AFTER hydra_overrides_extra: with apply_postprocessing they
    win; without it the caller's values (on top of the constructor defaults) stand."""
    from det_sam2_amd.build_sam import resolve_build_cfg
    extras = ["++model.sam_mask_decoder_extra_args.dynamic_multimask_via_stability=false", "++model.fill_hole_area=0",
              "++model.sam_mask_decoder_extra_args.dynamic_multimask_stability_thresh=0.9", "model.max_cond_frames_in_attn=4"]
    on = resolve_build_cfg("sam2.1_hiera_t", extras, True)
    assert (on.dynamic_multimask_via_stability, on.fill_hole_area, on.dynamic_multimask_stability_thresh, on.max_cond_frames_in_attn) == (True, 8, 0.98, 4)
    off = resolve_build_cfg("sam2.1_hiera_t", extras, False)
    assert (off.dynamic_multimask_via_stability, off.fill_hole_area, off.dynamic_multimask_stability_thresh, off.binarize_mask_from_pts_for_mem_enc) == (False, 0, 0.9, False)
