"""GPU parity AT THE MEASURED SHAPE and on BASELINE config 3's model (VERDICT r3 missing #2), against goldens written by the
REFERENCE itself (oracle/make_goldens.py e2e_large_b16 / e2e_bplus and their held-out `s1` variants):

* ``e2e_large_b16`` - sam2.1_hiera_l, 16 objects, one reverse pass over 9 frames: the last tracked frames attend 1 conditioning
  + 6 non-conditioning frames + object pointers, i.e. bench.py's Nk = 28 736 bank, through VideoProcessor.process_frame;
* ``e2e_bplus`` - sam2.1_hiera_base_plus, 4 objects, a preloaded bank of P = 1 conditioning frame written as a DS2BANK file,
  4 frames tracked with detect_interval = -1 (BASELINE config 3's scenario at test size).

Bar (BASELINE.json): 1 - IoU <= 1e-3 per (frame, object); logits within DLOGIT_TOL (2 x the measured difference) of the fixture's."""
import os

import numpy as np
import pytest

from _util import record
from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
LARGE, BPLUS = "sam2.1_hiera_l", "sam2.1_hiera_b+"
# max |dlogit| per fixture and mode: 2 x the value measured on MI355X (the kernels are deterministic) - hiera_l re-measured in round 6
# with the MX product in Hiera stages 3 / 4 (profiles/r06_final_metrics.jsonl: 1.0e-3 / 1.5e-3 in mode bf16x3, 1.7e-3, 5.2e-5, 3.3e-3),
# hiera_b+ as in round 5 (no MX layer at its widths)
DLOGIT_TOL = {("large", "seed0"): 3.0e-3, ("large", "s1"): 3.4e-3, ("large", "lm"): 1.1e-4, ("large", "s2"): 6.6e-3,
              ("bplus", "seed0"): 7.7e-3, ("bplus", "seed0", "fp32"): 8e-4, ("bplus", "s1"): 4.1e-3, ("bplus", "s2"): 1.5e-2}


def _iou(a, b):
    inter, union = np.logical_and(a, b).sum(), np.logical_or(a, b).sum()
    return 1.0 if union == 0 else inter / union


def _variant(v):
    from oracle.make_goldens import HELDOUT
    return (0, 1.0, False) if v == "seed0" else HELDOUT[v]


def _predictor(name, variant, prec, max_batch):
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    ws, ls, st = _variant(variant)
    cfg = resolve_config(name)
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, ws, ls), "cuda:0", max_batch=max_batch)
    pred.hip.set_precision(prec)
    return pred, st


@pytest.mark.parametrize("variant,prec", [("seed0", "bf16x3k"), ("seed0", "bf16x3"), ("s1", "bf16x3k"), ("lm", "bf16x3k"), ("s2", "bf16x3k")])
def test_hiera_large_16_objects_full_bank(golden_dir, variant, prec):
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from oracle.make_goldens import L16_FRAMES, L16_KW
    g = np.load(os.path.join(golden_dir, "e2e_large_b16.npz" if variant == "seed0" else f"ho_large_b16_{variant}.npz"))
    pred, st = _predictor(LARGE, variant, prec, 16)
    vp = VideoProcessor(model_cfg=LARGE, detector=SyntheticDetector(16), predictor=pred, **L16_KW)
    lows = []
    pred.trace = []
    orig = vp.predictor.propagate_in_video

    def capture(state, **k):
        for t, ids, bits in orig(state, **k):
            od = state["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            lows.append((t, len(ids), od[key][t]["pred_masks"].clone()))
            yield t, ids, bits

    vp.predictor.propagate_in_video = capture
    for t in range(L16_FRAMES):
        vp.process_frame(t, synthetic_frame(t, structured=st))
    assert [l[0] for l in lows] == list(g["frames"]) and [l[1] for l in lows] == list(g["nobj"])
    amax = float(g["logit_absmax"])
    worst, worst_low, dlogit, per_frame = 0.0, 0.0, 0.0, []
    for i, (t, nobj, low) in enumerate(lows):
        low = low.cpu().numpy()
        # the bar: IoU of the VIDEO-RESOLUTION masks (what VideoProcessor hands out; BASELINE.json "mask IoU vs ref"), in full
        seg = np.stack([vp.video_segments[t][oid] for oid in vp.inference_state["obj_ids"][:nobj]])
        rb = np.unpackbits(g[f"bitsfull{i}"])[: seg.size].reshape(seg.shape).astype(bool)
        w = max(1.0 - _iou(seg[o], rb[o]) for o in range(nobj))
        per_frame.append(w)
        worst = max(worst, w)
        # diagnostics: the sign pattern of the 256 x 256 low-res logits (a 16x smaller mask: one pixel of a ~5 500-pixel mask is
        # 2e-4) and the logits themselves
        ref_bits = np.unpackbits(g[f"lowbits{i}"])[: low.size].reshape(low.shape).astype(bool)
        worst_low = max(worst_low, max(1.0 - _iou(low[o] > 0, ref_bits[o]) for o in range(nobj)))
        sub = low[:, :, ::4, ::4]
        ref = g[f"low{i}"].astype(np.float32)                       # fp16 storage of the fixture: 2^-11 relative
        dlogit = max(dlogit, float((np.abs(sub - ref) - np.abs(ref) * 2.0 ** -11).max()))
    record("e2e_large_b16", variant=variant, prec=prec, one_minus_iou=worst, one_minus_iou_lowres=worst_low, max_abs_dlogit=dlogit,
           logit_absmax=amax, per_frame=[float(x) for x in per_frame])
    assert worst <= 1e-3 and worst_low <= 1e-3 and dlogit <= DLOGIT_TOL[("large", variant)], (worst, worst_low, dlogit, amax, per_frame)
    # the pass reaches the bench's bank: frames 2 and 1 attend 1 conditioning + 6 non-conditioning frames
    nks = [tr["nk"] for tr in pred.trace]
    assert max(nks) >= 4096 * 7 and min(nks) == 4096, nks


@pytest.mark.parametrize("variant,prec", [("seed0", "bf16x3k"), ("seed0", "bf16x3"), ("seed0", "fp32"), ("s1", "bf16x3k"), ("s2", "bf16x3k")])
def test_hiera_base_plus_preloaded_bank(golden_dir, tmp_path, variant, prec):
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from oracle.make_goldens import BPLUS_A, BPLUS_B, BPLUS_OBJECTS
    g = np.load(os.path.join(golden_dir, "e2e_bplus.npz" if variant == "seed0" else f"ho_bplus_{variant}.npz"))
    bank = str(tmp_path / "bank_bplus.ds2")
    strip = lambda kw: {k: v for k, v in kw.items() if k != "skip_classes"}  # noqa: E731
    pa, st = _predictor(BPLUS, variant, prec, BPLUS_OBJECTS)
    a = VideoProcessor(model_cfg=BPLUS, detector=SyntheticDetector(BPLUS_OBJECTS), skip_classes=set(), predictor=pa,
                       save_inference_state_path=bank, **strip(BPLUS_A))
    a.run(frames=[synthetic_frame(0, structured=st)])
    assert os.path.getsize(bank) > 0
    del a, pa
    pb, _ = _predictor(BPLUS, variant, prec, BPLUS_OBJECTS)
    b = VideoProcessor(model_cfg=BPLUS, detector=SyntheticDetector(BPLUS_OBJECTS), skip_classes=set(), predictor=pb,
                       load_inference_state_path=bank, **strip(BPLUS_B))
    segs = b.run(frames=[synthetic_frame(100 + i, structured=st) for i in range(4)])
    assert b.pre_frames == 1 and sorted(segs) == [0, 1, 2, 3]
    assert b.pass_log[0][1] == list(g["frames"])
    od = b.inference_state["output_dict"]
    worst, dlogit = 0.0, 0.0
    for i, t in enumerate(g["frames"]):
        low = od["non_cond_frame_outputs"][int(t)]["pred_masks"].cpu().numpy()
        dlogit = max(dlogit, float(np.abs(low - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(BPLUS_OBJECTS, 1, 1024, 1024).astype(bool)
        for o in range(BPLUS_OBJECTS):
            worst = max(worst, 1.0 - _iou(segs[int(t) - 1][o], ref[o]))
    amax = float(np.abs(g["low"]).max())
    record("e2e_bplus", variant=variant, prec=prec, one_minus_iou=worst, max_abs_dlogit=dlogit, logit_absmax=amax)
    assert worst <= 1e-3 and dlogit <= DLOGIT_TOL.get(("bplus", variant, prec), DLOGIT_TOL[("bplus", variant)]), (worst, dlogit, amax)


def test_hole_filling_default_at_measured_shape(golden_dir):
    """The SHIPPING default at the measured shape (VERDICT r5 missing #3): the predictor as build_sam2_video_predictor builds it
    (build_sam.py:126-135 appends fill_hole_area = 8; sam2_video_predictor.py:1343-1346 runs fill_holes_in_mask_scores on every
    inferred frame), hiera_l x 16 objects, one reverse pass over 9 frames up to the bench's bank.  The expectation is the ORACLE's
    (oracle/make_oracle_fixtures.py fill8_large_b16 -> oracle_fill8_large_b16.npz): the CPU reference cannot fill - its connected-
    components kernel is CUDA-only and misc.py:389-391 then returns the input - so there is no reference golden of this path.
    The uniform-noise frames make it a hard case: the filling moves 13 % of the low-res pixels (1.25 M of 9.4 M), and one flipped
    logit next to a background component of 8 or 9 pixels moves up to 9 more."""
    from det_sam2_amd.build_sam import build_sam2_video_predictor
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from oracle.make_goldens import L16_FRAMES, L16_KW
    g = np.load(os.path.join(golden_dir, "oracle_fill8_large_b16.npz"))
    cfg = resolve_config(LARGE)
    pred = build_sam2_video_predictor("configs/sam2.1/sam2.1_hiera_l.yaml", {"model": synthetic_state_dict(cfg, 0)}, device="cuda:0", max_batch=16)
    assert pred.fill_hole_area == 8
    vp = VideoProcessor(model_cfg=LARGE, detector=SyntheticDetector(16), predictor=pred, **L16_KW)
    for t in range(L16_FRAMES):
        vp.process_frame(t, synthetic_frame(t))
    assert vp.pass_log[0][1] == list(g["frames"])
    od = vp.inference_state["output_dict"]
    worst, worst_low, flips = 0.0, 0.0, 0
    for i, t in enumerate(g["frames"]):
        t = int(t)
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][t]["pred_masks"].cpu().numpy()
        ref_low = np.unpackbits(g[f"lowbits{i}"])[: low.size].reshape(low.shape).astype(bool)
        flips += int(((low > 0) != ref_low).sum())
        seg = np.stack([vp.video_segments[t][o] for o in range(16)])
        ref = np.unpackbits(g[f"bitsfull{i}"])[: seg.size].reshape(seg.shape).astype(bool)
        for o in range(16):
            worst = max(worst, 1.0 - _iou(seg[o], ref[o]))
            worst_low = max(worst_low, 1.0 - _iou(low[o] > 0, ref_low[o]))
    record("fill8_large_b16", one_minus_iou=worst, one_minus_iou_lowres=worst_low, lowres_sign_flips=flips,
           filled_lowres_pixels=int(g["filled_lowres_pixels"]))
    assert worst <= 1e-3 and worst_low <= 1e-3, (worst, worst_low, flips)
