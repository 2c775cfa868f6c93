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


@pytest.mark.parametrize("prec", ["fp32", "bf16x3k"])
def test_prompt_and_object_removal_match_reference(golden_dir, prec):
    """clear_all_prompts_in_frame / remove_object (sam2_video_predictor.py:1061-1131, :1438-1549) through the C-ABI against
    the reference golden e2e_remove (oracle/make_goldens.py): 3 objects + a second conditioning frame that only object 1
    has an input on; a correction click that is cleared again; object 1 removed (frame 3 demoted, row 1 cut out of every
    stored entry); propagation with the 2 remaining objects.  Bar: 1 - IoU <= 1e-3 per (frame, object)."""
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    from oracle.make_goldens import correction_prompts, removal_click
    cfg = resolve_config("sam2.1_hiera_t")
    g = np.load(os.path.join(golden_dir, "e2e_remove.npz"))
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=3)
    pred.hip.set_precision(prec)

    def worst_of(vr, packed, n):
        ref = np.unpackbits(packed).reshape(n, 1, 1024, 1024).astype(bool)
        got = (vr > 0).cpu().numpy()
        return max(1.0 - _iou(got[o], ref[o]) for o in range(n))

    st = pred.init_state([synthetic_frame(t) for t in range(6)])
    for o in range(3):
        pred.add_new_points_or_box(st, 0, o, box=synthetic_box(o, 0))
    pts, lab = removal_click()
    pred.add_new_points_or_box(st, 3, 1, points=pts, labels=lab)
    worst, worst_logit = 0.0, 0.0
    for i, (t, ids, logits) in enumerate(pred.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6)):
        assert t == g["first_frames"][i]
        worst = max(worst, worst_of(logits, g["first_bits"][i], 3))
    od = st["output_dict"]
    assert sorted(od["cond_frame_outputs"]) == list(g["first_cond"])
    c0 = correction_prompts()[0]
    _, _, vr = pred.add_new_points_or_box(st, 4, 0, points=c0[3], labels=c0[4])
    worst = max(worst, worst_of(vr, g["click_bits"], 3))
    t, ids, vr = pred.clear_all_prompts_in_frame(st, 4, 0)
    assert t == 4 and list(ids) == [0, 1, 2]
    worst = max(worst, worst_of(vr, g["clear_bits"], 3))
    ids, updated = pred.remove_object(st, 1)
    assert list(ids) == list(g["ids_after"]) and sorted(t for t, _ in updated) == sorted(g["updated_frames"])
    for t, vr in updated:
        worst = max(worst, worst_of(vr, g["updated_bits"][list(g["updated_frames"]).index(t)], 2))
    assert sorted(od["cond_frame_outputs"]) == list(g["cond_after"])
    assert sorted(od["non_cond_frame_outputs"]) == list(g["noncond_after"])
    assert sorted(st["frames_already_tracked"]) == list(g["tracked_after"])
    assert sorted(st["consolidated_frame_inds"]["cond_frame_outputs"]) == list(g["consolidated_cond_after"])
    worst_logit = float(np.abs(od["non_cond_frame_outputs"][3]["pred_masks"].cpu().numpy() - g["low3_after"]).max())
    ids2, upd2 = pred.remove_object(st, 77)
    assert list(ids2) == list(ids) and upd2 == []
    with pytest.raises(RuntimeError):
        pred.remove_object(st, 77, strict=True)
    ys = [(t, lg.clone()) for t, ids, lg in pred.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6)]
    assert [t for t, _ in ys] == list(g["frames"])
    for i, (t, lg) in enumerate(ys):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        assert od[key][t]["pred_masks"].shape[0] == 2
        worst_logit = max(worst_logit, float(np.abs(od[key][t]["pred_masks"].cpu().numpy() - g["low"][i]).max()))
        worst = max(worst, worst_of(lg, g["bits"][i], 2))
    record("e2e_remove", prec=prec, one_minus_iou=worst, max_abs_dlogit=worst_logit, logit_absmax=float(np.abs(g["low"]).max()))
    assert worst <= 1e-3 and worst_logit <= (5e-3 if prec == "fp32" else 5e-2), (worst, worst_logit)
