"""BASELINE configs 2 / 5 at reduced size on the GPU (tiny model): size-independent properties of the streaming
driver with Det-SAM2's default schedule (30 / 30 / 60 / 60) - where the oracle would take hours.

* every frame ends up with a mask for every object known at that time; a 5th class appearing mid-stream (A17) is
  tracked from its first pass on;
* the pass schedule is the reference's (reverse passes from 30k+29 over 60 frames, det_sam2_RT.py:388-393,429);
* eviction (release_old_frames, release_images=True) bounds the state: retained images / bank entries / peak HBM stay
  flat once the window is full ("assert peak VRAM flat after frame 120");
* determinism: the same stream run twice gives bit-identical masks (no atomics-order or race dependence)."""
import numpy as np
import pytest
import torch

from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

from _util import record

pytestmark = pytest.mark.gpu
TINY = "sam2.1_hiera_t"
N = 240


def _run(n_frames, checkpoints=()):
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config(TINY)
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=8)
    vp = VideoProcessor(model_cfg=TINY, detector=SyntheticDetector(5, appear={4: 90}), skip_classes=set(), predictor=pred)
    assert (vp.frame_buffer_size, vp.detect_interval, vp.max_frame_num_to_track, vp.max_inference_state_frames) == (30, 30, 60, 60)
    mem = {}
    for t in range(n_frames):
        vp.process_frame(t, synthetic_frame(t))
        if t in checkpoints:
            torch.cuda.synchronize()
            st = vp.inference_state
            mem[t] = dict(peak=torch.cuda.max_memory_allocated(), alloc=torch.cuda.memory_allocated(),
                          images=len(st["images_idx"]), cond=len(st["output_dict"]["cond_frame_outputs"]),
                          noncond=len(st["output_dict"]["non_cond_frame_outputs"]), cached=len(st["cached_features"]))
            torch.cuda.reset_peak_memory_stats()
    return vp, mem


def test_default_schedule_long_stream_properties():
    vp, mem = _run(N, checkpoints=(119, 149, 179, 209, 239))
    # pass schedule
    assert [p[0] for p in vp.pass_log] == [30 * k + 29 for k in range(N // 30)]
    assert vp.pass_log[0][1] == list(range(29, -1, -1))
    for k in range(1, N // 30):
        assert vp.pass_log[k][1] == list(range(30 * k + 29, 30 * k - 31, -1))
    # coverage: all frames, 4 objects before the 5th class is first detected (frame 90), 5 in every pass after
    assert sorted(vp.video_segments) == list(range(N))
    for t in range(N):
        ids = sorted(vp.video_segments[t])
        assert ids == ([0, 1, 2, 3] if t < 60 else [0, 1, 2, 3, 4]), (t, ids)   # pass 3 (frames 60..119) is the first with class 4
        for m in vp.video_segments[t].values():
            assert m.shape == (1, 1024, 1024) and m.dtype == bool
    # bounded state
    for t, s in mem.items():
        assert s["images"] <= 90 and s["noncond"] <= 90 and s["cond"] <= 4 and s["cached"] <= 90, (t, s)
    base = mem[149]
    for t in (179, 209, 239):
        assert mem[t]["images"] == base["images"] and mem[t]["noncond"] == base["noncond"] and mem[t]["cond"] == base["cond"]
        assert mem[t]["alloc"] <= base["alloc"] * 1.02 + (8 << 20), (t, mem[t], base)
        assert mem[t]["peak"] <= base["peak"] * 1.02 + (8 << 20), (t, mem[t], base)
    record("longrun", frames=N, peak_mib=base["peak"] / 2 ** 20, alloc_mib=base["alloc"] / 2 ** 20, images=base["images"],
           noncond=base["noncond"], tracked=vp.predictor.stats["tracked_frames"])


def test_stream_is_deterministic():
    a, _ = _run(90)
    b, _ = _run(90)
    diff = 0
    for t in range(90):
        for o in a.video_segments[t]:
            diff += int((a.video_segments[t][o] != b.video_segments[t][o]).sum())
    assert diff == 0


def test_async_encode_equals_synchronous(monkeypatch):
    """The encoder batch launched ahead on the second stream (second model instance, HIP-event hand-off) must give the
    very same masks as encoding on the caller's stream: 90 frames of the default schedule (three passes, every frame
    tracked twice, eviction not yet active), compared bit for bit."""
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config(TINY)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DS2_ASYNC_ENCODE", mode)
        pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=8)
        assert pred.async_encode == (mode == "1")
        vp = VideoProcessor(model_cfg=TINY, detector=SyntheticDetector(4), skip_classes=set(), predictor=pred)
        for t in range(90):
            vp.process_frame(t, synthetic_frame(t))
        torch.cuda.synchronize()
        out[mode] = vp.video_segments
        if mode == "1":
            assert pred._hip_enc is not None, "the async path never ran"      # (the test would be vacuous)
        del vp, pred
    assert sorted(out["1"]) == sorted(out["0"]) == list(range(90))
    differing = sum(int((np.asarray(out["1"][t][o]) != np.asarray(out["0"][t][o])).sum()) for t in range(90) for o in out["0"][t])
    assert differing == 0, differing
