"""End-to-end GPU parity: the MI355X VideoProcessor/SAM2VideoPredictor (HIP stages through the C-ABI)
against golden vectors produced by the REFERENCE itself (tests/golden, oracle/make_goldens.py).
Bar (BASELINE.json): 1 - IoU <= 1e-3 per (frame, object) on identical frames; logits within fp32 noise."""
import os

import numpy as np
import pytest
import torch

from _util import record
from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
TINY = "sam2.1_hiera_t"


def _iou(a, b):
    inter, union = np.logical_and(a, b).sum(), np.logical_or(a, b).sum()
    return 1.0 if union == 0 else inter / union


@pytest.fixture(params=["fp32", "bf16x3", "bf16x3k"])
def prec(request):
    return request.param


def _vp(detector, prec, **kw):
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config(TINY)
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=4)
    pred.hip.set_precision(prec)
    return VideoProcessor(model_cfg=TINY, detector=detector, skip_classes=set(), predictor=pred, **kw)


def test_config1_matches_reference(golden_dir, prec):
    g = np.load(os.path.join(golden_dir, "e2e_cfg1.npz"))
    vp = _vp(SyntheticDetector(1), prec, frame_buffer_size=8, detect_interval=8, max_frame_num_to_track=8, max_inference_state_frames=-1)
    for t in range(8):
        vp.process_frame(t, synthetic_frame(t))
    assert vp.pass_log[0][1] == list(g["frames"])
    od = vp.inference_state["output_dict"]
    worst_iou, worst_logit = 0.0, 0.0
    for i, t in enumerate(g["frames"]):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][int(t)]["pred_masks"].cpu().numpy()
        worst_logit = max(worst_logit, float(np.abs(low - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(1, 1024, 1024).astype(bool)
        worst_iou = max(worst_iou, 1.0 - _iou(vp.video_segments[int(t)][0], ref))
    record("e2e_cfg1", prec=prec, one_minus_iou=worst_iou, max_abs_dlogit=worst_logit, logit_absmax=float(np.abs(g["low"]).max()))
    assert worst_iou <= 1e-3 and worst_logit <= (5e-3 if prec == "fp32" else 5e-2), (worst_iou, worst_logit)
    assert vp.predictor.stats["encoder_runs"] == 8


def test_duplicate_class_matches_reference(golden_dir, prec):
    """Two boxes of one class on the prompted frame => second prompt with the first prediction as mask prompt."""
    g = np.load(os.path.join(golden_dir, "e2e_dup.npz"))
    vp = _vp(SyntheticDetector(2, duplicates={0: 1}), prec, frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=4,
             max_inference_state_frames=-1)
    for t in range(4):
        vp.process_frame(t, synthetic_frame(t))
    assert vp.pass_log[0][1] == list(g["frames"])
    od = vp.inference_state["output_dict"]
    worst_iou, worst_logit = 0.0, 0.0
    for i, t in enumerate(g["frames"]):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][int(t)]["pred_masks"].cpu().numpy()
        worst_logit = max(worst_logit, float(np.abs(low - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(2, 1, 1024, 1024).astype(bool)
        for o in range(2):
            worst_iou = max(worst_iou, 1.0 - _iou(vp.video_segments[int(t)][o], ref[o]))
    record("e2e_dup", prec=prec, one_minus_iou=worst_iou, max_abs_dlogit=worst_logit)
    assert worst_iou <= 1e-3 and worst_logit <= (5e-3 if prec == "fp32" else 5e-2), (worst_iou, worst_logit)


def test_preload_bank_matches_reference(golden_dir, prec, tmp_path):
    """A18: build a bank, pickle it (save_inference_state), preload it in a fresh VideoProcessor.run and track 4 new
    frames without a detector; masks vs the reference's own preload run."""
    from oracle.make_goldens import PRELOAD_A, PRELOAD_B
    g = np.load(os.path.join(golden_dir, "e2e_preload.npz"))
    bank = str(tmp_path / "bank.pkl")
    a = _vp(SyntheticDetector(2), prec, save_inference_state_path=bank, **{k: v for k, v in PRELOAD_A.items() if k != "skip_classes"})
    a.run(frames=[synthetic_frame(t) for t in range(3)])
    assert os.path.getsize(bank) > 0
    b = _vp(SyntheticDetector(2), prec, load_inference_state_path=bank, **{k: v for k, v in PRELOAD_B.items() if k != "skip_classes"})
    segs = b.run(frames=[synthetic_frame(100 + i) for i in range(4)])
    assert b.pre_frames == 3 and sorted(segs) == [0, 1, 2, 3]          # run() re-bases frame indices (:612)
    assert b.pass_log[0][1] == list(g["frames"])
    od = b.inference_state["output_dict"]
    worst_iou, worst_logit = 0.0, 0.0
    for i, t in enumerate(g["frames"]):
        low = od["non_cond_frame_outputs"][int(t)]["pred_masks"].cpu().numpy()
        worst_logit = max(worst_logit, float(np.abs(low - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(2, 1, 1024, 1024).astype(bool)
        for o in range(2):
            worst_iou = max(worst_iou, 1.0 - _iou(segs[int(t) - 3][o], ref[o]))
    record("e2e_preload", prec=prec, one_minus_iou=worst_iou, max_abs_dlogit=worst_logit)
    assert worst_iou <= 1e-3 and worst_logit <= (5e-3 if prec == "fp32" else 5e-2), (worst_iou, worst_logit)


def test_preload_bank_written_by_the_reference_layout(golden_dir, tmp_path):
    """A bank in the REFERENCE's on-disk form - pickle of the whole state with channel-major maskmem_features
    [B,64,64,64], maskmem_pos_enc lists and the frames (det_sam2_RT.py:489-497), produced here by the oracle's
    VideoProcessor - is loaded through load_inference_state_path, converted, and tracks the 4 new frames to the masks of
    the reference's own preload run (golden e2e_preload)."""
    import torch
    from oracle.make_goldens import PRELOAD_A, PRELOAD_B
    from oracle.video_processor import OracleVideoProcessor
    g = np.load(os.path.join(golden_dir, "e2e_preload.npz"))
    cfg = resolve_config(TINY)
    bank = str(tmp_path / "reference_layout_bank.pkl")
    a = OracleVideoProcessor(synthetic_state_dict(cfg, 0), cfg, SyntheticDetector(2), **PRELOAD_A)
    with torch.inference_mode():
        for t in range(3):
            a.process_frame(t, synthetic_frame(t))
    a.save_inference_state(bank)
    b = _vp(SyntheticDetector(2), "bf16x3k", load_inference_state_path=bank, **{k: v for k, v in PRELOAD_B.items() if k != "skip_classes"})
    segs = b.run(frames=[synthetic_frame(100 + i) for i in range(4)])
    assert b.pre_frames == 3 and sorted(segs) == [0, 1, 2, 3]
    assert b.pass_log[0][1] == list(g["frames"])
    worst = 0.0
    for i, t in enumerate(g["frames"]):
        ref = np.unpackbits(g["bits"][i]).reshape(2, 1, 1024, 1024).astype(bool)
        for o in range(2):
            worst = max(worst, 1.0 - _iou(segs[int(t) - 3][o], ref[o]))
    record("e2e_preload_reference_layout", one_minus_iou=worst)
    assert worst <= 1e-3, worst


def test_stream2_matches_reference(golden_dir, prec):
    """Two passes, release_old_frames with image release, online new object (A17)."""
    g = np.load(os.path.join(golden_dir, "e2e_stream2.npz"))
    vp = _vp(SyntheticDetector(3, appear={2: 4}), prec, frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=8,
             max_inference_state_frames=6)
    lows = []
    orig = vp.predictor.propagate_in_video

    def capture(st, **kw):
        for t, ids, bits in orig(st, **kw):
            od = st["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            lows.append((t, len(ids), od[key][t]["pred_masks"].clone()))
            yield t, ids, bits

    vp.predictor.propagate_in_video = capture
    for t in range(8):
        vp.process_frame(t, synthetic_frame(t))
    assert [l[0] for l in lows] == list(g["frames"])
    assert [l[1] for l in lows] == list(g["nobj"])
    st = vp.inference_state
    assert sorted(st["output_dict"]["cond_frame_outputs"]) == list(g["final_cond"])
    assert sorted(st["output_dict"]["non_cond_frame_outputs"]) == list(g["final_noncond"])
    assert st["images_idx"] == list(g["images_idx"])
    worst = 0.0
    for i, (t, nobj, low) in enumerate(lows):
        low = low.cpu().numpy()
        ref_bits = np.unpackbits(g[f"lowbits{i}"])[: low.size].reshape(low.shape).astype(bool)
        for o in range(nobj):
            worst = max(worst, 1.0 - _iou(low[o] > 0, ref_bits[o]))
        assert np.abs(low - g[f"low{i}"].astype(np.float32)).max() <= 2e-2 + 1e-3 * np.abs(low).max()
    # final masks of the second pass at video resolution (2x decimated in the fixture)
    n_first = int((g["pass_id"] == 0).sum())
    for i in range(n_first, len(lows)):
        t, nobj, _ = lows[i]
        seg = np.stack([vp.video_segments[t][oid] for oid in sorted(vp.video_segments[t])])[:, :, ::2, ::2]
        ref = np.unpackbits(g[f"bits{i}"])[: seg.size].reshape(seg.shape).astype(bool)
        for o in range(nobj):
            worst = max(worst, 1.0 - _iou(seg[o], ref[o]))
    record("e2e_stream2", prec=prec, one_minus_iou=worst)
    assert worst <= 1e-3, worst


def test_hiera_large_matches_reference(golden_dir):
    """The bench configuration's model (sam2.1_hiera_l) end to end against a golden produced by the REFERENCE itself
    (oracle/make_goldens.py e2e_large: 3 frames, 2 objects), default bf16x3k arithmetic; bar 1 - IoU <= 1e-3 per mask."""
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    from oracle.make_goldens import LARGE_KW
    name = "sam2.1_hiera_l"
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, 0)
    g = np.load(os.path.join(golden_dir, "e2e_large.npz"))
    pred = SAM2VideoPredictor(cfg, sd, "cuda:0", max_batch=2)
    pred.hip.set_precision("bf16x3k")         # the default (shipping) arithmetic mode
    vp = VideoProcessor(model_cfg=name, detector=SyntheticDetector(2), predictor=pred, **LARGE_KW)
    for t in range(3):
        vp.process_frame(t, synthetic_frame(t))
    assert vp.pass_log[0][1] == list(g["frames"])
    worst, worst_logit = 0.0, 0.0
    od = vp.inference_state["output_dict"]
    for i, t in enumerate(g["frames"]):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][int(t)]["pred_masks"].cpu().numpy()
        worst_logit = max(worst_logit, float(np.abs(low - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(2, 1, 1024, 1024).astype(bool)
        for o in range(2):
            worst = max(worst, 1.0 - _iou(vp.video_segments[int(t)][o], ref[o]))
    record("e2e_hiera_l_ref", one_minus_iou=worst, max_abs_dlogit=worst_logit, logit_absmax=float(np.abs(g["low"]).max()))
    assert worst <= 1e-3 and worst_logit <= 5e-2, (worst, worst_logit)


def _run_compact(golden, detector, kw, n_frames, max_batch, prec="bf16x3k"):
    """Drive the HIP VideoProcessor and compare every propagate yield with a _compact() reference fixture."""
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config(TINY)
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=max_batch)
    pred.hip.set_precision(prec)
    vp = VideoProcessor(model_cfg=TINY, detector=detector, predictor=pred, **kw)
    lows = []
    orig = vp.predictor.propagate_in_video

    def capture(st, **k):
        for t, ids, bits in orig(st, **k):
            od = st["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            lows.append((t, len(ids), od[key][t]["pred_masks"].clone()))
            yield t, ids, bits

    vp.predictor.propagate_in_video = capture
    for t in range(n_frames):
        vp.process_frame(t, synthetic_frame(t))
    g = golden
    assert [l[0] for l in lows] == list(g["frames"])
    assert [l[1] for l in lows] == list(g["nobj"])
    worst, worst_logit = 0.0, 0.0
    for i, (t, nobj, low) in enumerate(lows):
        low = low.cpu().numpy()
        ref_bits = np.unpackbits(g[f"lowbits{i}"])[: low.size].reshape(low.shape).astype(bool)
        for o in range(nobj):
            worst = max(worst, 1.0 - _iou(low[o] > 0, ref_bits[o]))
        sub = low[:, :, ::4, ::4]
        d = float(np.abs(sub - g[f"low{i}"].astype(np.float32)).max())
        worst_logit = max(worst_logit, d)
        assert d <= 5e-2 + 1e-3 * float(np.abs(sub).max()), (i, d)
    # video-res masks of the LAST pass covering each frame (what video_segments holds), 4x decimated in the fixture
    last = {}
    for i, (t, nobj, _) in enumerate(lows):
        last[t] = i
    for t, i in last.items():
        seg = np.stack([vp.video_segments[t][oid] for oid in vp.inference_state["obj_ids"][: lows[i][1]]])[:, :, ::4, ::4]
        ref = np.unpackbits(g[f"bits{i}"])[: seg.size].reshape(seg.shape).astype(bool)
        for o in range(lows[i][1]):
            worst = max(worst, 1.0 - _iou(seg[o], ref[o]))
    return vp, worst, worst_logit


def test_16_objects_matches_reference(golden_dir, prec):
    """B = 16 end to end (the batch of BASELINE configs 3-5) against the reference golden e2e_b16."""
    from oracle.make_goldens import B16_KW
    g = np.load(os.path.join(golden_dir, "e2e_b16.npz"))
    vp, worst, worst_logit = _run_compact(g, SyntheticDetector(16), B16_KW, 3, 16, prec)
    record("e2e_b16", prec=prec, one_minus_iou=worst, max_abs_dlogit=worst_logit)
    assert worst <= 1e-3, (worst, worst_logit)


def test_17th_object_online_matches_reference(golden_dir):
    """BASELINE config 5's mid-stream new category at full batch, against the reference golden e2e_b17: the model is
    created for max_batch = 16, the 17th class appears in the second pass => workspace growth, A17 re-consolidation of
    the cond frames to B = 17, and 17 x 16 = 272 cross-attention blocks (not a multiple of 8 per object)."""
    from oracle.make_goldens import B17_KW
    g = np.load(os.path.join(golden_dir, "e2e_b17.npz"))
    vp, worst, worst_logit = _run_compact(g, SyntheticDetector(17, appear={16: 2}), B17_KW, 4, 16)
    od = vp.inference_state["output_dict"]
    assert sorted(od["cond_frame_outputs"]) == list(g["final_cond"])
    assert sorted(od["non_cond_frame_outputs"]) == list(g["final_noncond"])
    record("e2e_b17", one_minus_iou=worst, max_abs_dlogit=worst_logit)
    assert worst <= 1e-3, (worst, worst_logit)


def test_classes_match_reference(golden_dir, prec):
    """A2 branches with the reference's DEFAULT skip_classes {11, 14, 15, 19}: class 14 is skipped, class 11 is
    collected in special_classes_detection (and, being in skip_classes, not tracked); golden e2e_classes."""
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    from oracle.make_goldens import CLASSES_IDS, CLASSES_KW
    g = np.load(os.path.join(golden_dir, "e2e_classes.npz"))
    cfg = resolve_config(TINY)
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=4)
    pred.hip.set_precision(prec)
    vp = VideoProcessor(model_cfg=TINY, detector=SyntheticDetector(5, class_ids=CLASSES_IDS, appear={4: 1}), predictor=pred, **CLASSES_KW)
    assert vp.skip_classes == {11, 14, 15, 19}          # the constructor default (det_sam2_RT.py:34)
    for t in range(2):
        vp.process_frame(t, synthetic_frame(t))
    assert list(vp.inference_state["obj_ids"]) == list(g["obj_ids"]) == [3, 7]
    assert vp.special_classes_count == int(g["special_count"]) == 2
    got = np.stack([np.asarray(b, np.float32).reshape(-1) for b in vp.special_classes_detection])
    assert np.array_equal(got, g["special"])
    od = vp.inference_state["output_dict"]
    worst, worst_logit = 0.0, 0.0
    for i, t in enumerate(g["frames"]):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][int(t)]["pred_masks"].cpu().numpy()
        worst_logit = max(worst_logit, float(np.abs(low - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(2, 1, 1024, 1024).astype(bool)
        for j, oid in enumerate(g["obj_ids"]):
            worst = max(worst, 1.0 - _iou(vp.video_segments[int(t)][int(oid)], ref[j]))
    record("e2e_classes", prec=prec, one_minus_iou=worst, max_abs_dlogit=worst_logit)
    assert worst <= 1e-3 and worst_logit <= (5e-3 if prec == "fp32" else 5e-2), (worst, worst_logit)


def test_three_pass_stream_matches_oracle(golden_dir):
    """Error accumulation through the memory bank: 12 frames, 3 overlapping reverse passes with eviction, 3 objects (one
    appearing in the second pass), default bf16x3k arithmetic, against the ORACLE's run of the same stream - its masks are the
    committed fixture oracle_three_pass.npz (oracle/make_oracle_fixtures.py three_pass; the run takes 1.5 minutes of host time, which
    this test spent on the GPU box until round 6; DS2_SLOW_ORACLE=1 runs the oracle alongside instead).  Every final mask within
    1 - IoU <= 1e-3."""
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    import torch
    cfg = resolve_config(TINY)
    sd = synthetic_state_dict(cfg, 0)
    kw = dict(skip_classes=set(), frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=8, max_inference_state_frames=8)
    det = lambda: SyntheticDetector(3, appear={2: 4})  # noqa: E731
    pred = SAM2VideoPredictor(cfg, sd, "cuda:0", max_batch=4)
    pred.hip.set_precision("bf16x3k")
    vp = VideoProcessor(model_cfg=TINY, detector=det(), predictor=pred, **kw)
    for t in range(12):
        vp.process_frame(t, synthetic_frame(t))
    if os.environ.get("DS2_SLOW_ORACLE"):
        from oracle.video_processor import OracleVideoProcessor
        ovp = OracleVideoProcessor(sd, cfg, det(), **kw)
        with torch.inference_mode():
            for t in range(12):
                ovp.process_frame(t, synthetic_frame(t))
        want_passes = [(p[0], list(p[1])) for p in ovp.pass_log]
        want = {t: {o: ovp.video_segments[t][o] for o in ovp.video_segments[t]} for t in range(12)}
    else:
        g = np.load(os.path.join(golden_dir, "oracle_three_pass.npz"))
        want_passes = [(int(s0), list(g[f"pass{i}"])) for i, s0 in enumerate(g["passes"])]
        want = {}
        for t in range(12):
            objs = [int(o) for o in g[f"objs{t}"]]
            bits = np.unpackbits(g[f"bits{t}"])[: len(objs) * 1024 * 1024].reshape(len(objs), 1, 1024, 1024).astype(bool)
            want[t] = {o: bits[j] for j, o in enumerate(objs)}
    assert [(p[0], list(p[1])) for p in vp.pass_log] == want_passes
    worst = 0.0
    for t in range(12):
        assert sorted(vp.video_segments[t]) == sorted(want[t]), t
        for o in vp.video_segments[t]:
            worst = max(worst, 1.0 - _iou(vp.video_segments[t][o], want[t][o]))
    record("e2e_three_pass", one_minus_iou=worst)
    assert worst <= 1e-3, worst


def test_without_postprocessing_matches_reference(golden_dir):
    """build_sam2_video_predictor(apply_postprocessing=False): token-0 single-mask output, sigmoid prompt masks into the memory
    encoder, no hole filling - against the reference built the same way (golden e2e_nopost)."""
    from det_sam2_amd.build_sam import build_sam2_video_predictor
    from det_sam2_amd.synth import synthetic_box
    g = np.load(os.path.join(golden_dir, "e2e_nopost.npz"))
    pred = build_sam2_video_predictor(TINY, None, device="cuda:0", apply_postprocessing=False, max_batch=2)
    assert not pred.cfg.dynamic_multimask_via_stability and pred.fill_hole_area == 0
    st = pred.init_state([synthetic_frame(t) for t in range(4)])
    for o in range(2):
        pred.add_new_points_or_box(st, 0, o, box=synthetic_box(o, 0))
    got = [(t, (lg > 0).cpu().numpy()) for t, _, lg in pred.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=4)]
    assert [t for t, _ in got] == list(g["frames"])
    od = st["output_dict"]
    worst, dlogit = 0.0, 0.0
    for i, (t, m) in enumerate(got):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][t]["pred_masks"].cpu().numpy()
        dlogit = max(dlogit, float(np.abs(low.reshape(g["low"][i].shape) - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(2, 1024, 1024).astype(bool)
        for o in range(2):
            worst = max(worst, 1.0 - _iou(m[o].reshape(1024, 1024), ref[o]))
    record("e2e_nopost", one_minus_iou=worst, max_abs_dlogit=dlogit)
    assert worst <= 1e-3 and dlogit <= 5e-2, (worst, dlogit)
