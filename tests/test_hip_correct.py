"""A6 completeness on the GPU: correction prompts on ALREADY-TRACKED frames through the C-ABI against the reference golden
e2e_correct (oracle/make_goldens.py: boxes on frame 0, propagate, a negative click / two clicks / a mask on tracked frames
2-4, propagate again).  Reference: sam2_video_predictor.py:428-483 (points), :583-586 (mask), preflight :836-857.
Bar: 1 - IoU <= 1e-3 per (frame, object)."""
import os

import numpy as np
import pytest

from _util import record
from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import synthetic_box, synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu


def _iou(a, b):
    inter, union = np.logical_and(a, b).sum(), np.logical_or(a, b).sum()
    return 1.0 if union == 0 else inter / union


@pytest.mark.parametrize("prec", ["fp32", "bf16x3k"])
def test_correction_prompts_match_reference(golden_dir, prec):
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    from oracle.make_goldens import correction_prompts
    cfg = resolve_config("sam2.1_hiera_t")
    g = np.load(os.path.join(golden_dir, "e2e_correct.npz"))
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=2)
    pred.hip.set_precision(prec)
    st = pred.init_state([synthetic_frame(t) for t in range(6)])
    for o in range(2):
        pred.add_new_points_or_box(st, 0, o, box=synthetic_box(o, 0))
    worst, worst_logit = 0.0, 0.0
    for i, (t, ids, logits) in enumerate(pred.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6)):
        assert t == g["first_frames"][i]
        ref = np.unpackbits(g["first_bits"][i]).reshape(2, 1, 1024, 1024).astype(bool)
        got = (logits > 0).cpu().numpy()
        for o in range(2):
            worst = max(worst, 1.0 - _iou(got[o], ref[o]))
    for i, (kind, t, oid, a, b) in enumerate(correction_prompts()):
        if kind == "points":
            _, ids, vr = pred.add_new_points_or_box(st, t, oid, points=a, labels=b)
        else:
            _, ids, vr = pred.add_new_mask(st, t, oid, a)
        tmp = st["temp_output_dict_per_obj"][oid]
        assert t in tmp["non_cond_frame_outputs"] and t not in tmp["cond_frame_outputs"]
        low = tmp["non_cond_frame_outputs"][t]["pred_masks"].cpu().numpy()
        worst_logit = max(worst_logit, float(np.abs(low - g[f"prompt_low{i}"]).max()))
        ref = np.unpackbits(g[f"prompt_bits{i}"]).reshape(2, 1, 1024, 1024).astype(bool)
        got = (vr > 0).cpu().numpy()
        for o in range(2):
            worst = max(worst, 1.0 - _iou(got[o], ref[o]))
    n0 = pred.stats["tracked_frames"]
    ys = [(t, (lg > 0).cpu().numpy()) for t, ids, lg in pred.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6)]
    assert pred.stats["tracked_frames"] == n0 + 2                      # frames 1 and 5; 2, 3, 4 are consolidated corrections
    assert [t for t, _ in ys] == list(g["frames"])
    od = st["output_dict"]
    assert sorted(od["cond_frame_outputs"]) == list(g["final_cond"])
    assert sorted(od["non_cond_frame_outputs"]) == list(g["final_noncond"])
    assert sorted(st["consolidated_frame_inds"]["non_cond_frame_outputs"]) == list(g["consolidated_noncond"])
    for i, (t, got) in enumerate(ys):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        worst_logit = max(worst_logit, float(np.abs(od[key][t]["pred_masks"].cpu().numpy() - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(2, 1, 1024, 1024).astype(bool)
        for o in range(2):
            worst = max(worst, 1.0 - _iou(got[o], ref[o]))
    record("e2e_correct", prec=prec, one_minus_iou=worst, max_abs_dlogit=worst_logit, logit_absmax=float(np.abs(g["low"]).max()))
    assert worst <= 1e-3 and worst_logit <= (5e-3 if prec == "fp32" else 5e-2), (worst, worst_logit)
