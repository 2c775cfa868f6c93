"""BASELINE configs 3, 4 and 5 at their REAL model / object-count sizes on the GPU (the oracle would need hours here, so
these are the size-independent properties; numerics at these sizes are pinned by the reference goldens e2e_large,
e2e_b16, e2e_b17 and the module fixtures of all four configs):

* config 3: sam2.1_hiera_base_plus, 16 objects, a PRELOADED bank of P conditioning frames (P = 1: the "7-frame bank";
  P = 10: the reference's example size, det_sam2_RT.py:660-665) written and read as a DS2BANK file, no detector
  afterwards (detect_interval = -1): every tracked frame attends Nk = 4096 (P + 6) + 64 keys in steady state;
* config 4 / 5: sam2.1_hiera_large, 16 objects, Det-SAM2's default 30/30/60/60 schedule over 180 frames with a 17th
  class first detected at stream frame 90 (A17 at full batch: workspace growth past max_batch = 16); the retained state
  and the HBM footprint are flat once the window is full ("assert peak VRAM flat after frame 120")."""
import numpy as np
import pytest
import torch

from _util import record
from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bplus():
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config("sam2.1_hiera_base_plus")
    return cfg, SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=16)


@pytest.mark.parametrize("P", [1, 10])
def test_config3_preloaded_bank_bplus_16_objects(bplus, P, tmp_path):
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    cfg, pred = bplus
    bank = str(tmp_path / f"bank_P{P}.ds2")
    a = VideoProcessor(model_cfg=cfg.name, detector=SyntheticDetector(16), skip_classes=set(), predictor=pred, frame_buffer_size=P,
                       detect_interval=1, max_frame_num_to_track=P, max_inference_state_frames=-1, save_inference_state_path=bank)
    a.run(frames=[synthetic_frame(t) for t in range(P)])
    assert len(a.inference_state["output_dict"]["cond_frame_outputs"]) == P
    n = 45
    b = VideoProcessor(model_cfg=cfg.name, detector=SyntheticDetector(16), skip_classes=set(), predictor=pred, frame_buffer_size=15,
                       detect_interval=-1, max_frame_num_to_track=30, max_inference_state_frames=30, load_inference_state_path=bank)
    pred.trace = []
    segs = b.run(frames=[synthetic_frame(200 + t) for t in range(n)])
    trace, pred.trace = pred.trace, None
    assert b.pre_frames == P and len(b.inference_state["images"]) <= 30            # the bank file carries no frames
    assert sorted(segs) == list(range(n))
    for t in range(n):
        assert sorted(segs[t]) == list(range(16)) and segs[t][3].shape == (1, 1024, 1024)
    # steady state: P preload conditioning frames + 6 non-conditioning ones + the pointers of the 15 later frames of the
    # reverse pass (the preload frames lie in the "future" of a reverse pass: their pointers are filtered, sam2_base.py:591-598)
    nk_full = 4096 * (P + 6) + 4 * 15
    nks = [tr["nk"] for tr in trace]
    assert max(nks) == nk_full and nks.count(nk_full) >= len(nks) // 3, (nk_full, sorted(set(nks)))
    assert all(len(tr["mem"]) <= P + 6 and [m for m in tr["mem"] if m[0] == 0] == [(0, t) for t in range(P)] for tr in trace)
    record("config3_bplus_preload", P=P, nk=nk_full, tracked=len(trace))


def test_config4_5_large_16_objects_default_schedule_17th_class_flat_vram():
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config("sam2.1_hiera_large")
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=16)
    vp = VideoProcessor(model_cfg=cfg.name, detector=SyntheticDetector(17, appear={16: 90}), skip_classes=set(), predictor=pred)
    assert (vp.frame_buffer_size, vp.detect_interval, vp.max_frame_num_to_track, vp.max_inference_state_frames) == (30, 30, 60, 60)
    n, mem = 180, {}
    for t in range(n):
        vp.process_frame(t, synthetic_frame(t))
        if t in (119, 149, 179):
            torch.cuda.synchronize()
            st = vp.inference_state
            mem[t] = dict(peak=torch.cuda.max_memory_allocated(), alloc=torch.cuda.memory_allocated(), images=len(st["images_idx"]),
                          cond=len(st["output_dict"]["cond_frame_outputs"]), noncond=len(st["output_dict"]["non_cond_frame_outputs"]),
                          cached=len(st["cached_features"]))
            torch.cuda.reset_peak_memory_stats()
    assert sorted(vp.video_segments) == list(range(n))
    for t in range(n):                      # pass 3 (frames 60..119) is the first one that knows the 17th class
        assert sorted(vp.video_segments[t]) == (list(range(16)) if t < 60 else list(range(17))), t
    assert vp.inference_state["output_dict"]["cond_frame_outputs"][150]["obj_ptr"].shape[0] == 17
    for t, s in mem.items():
        assert s["images"] <= 90 and s["noncond"] <= 90 and s["cond"] <= 4 and s["cached"] <= 90, (t, s)
    base = mem[149]                          # 17 objects from here on, window full
    assert mem[179]["images"] == base["images"] and mem[179]["noncond"] == base["noncond"] and mem[179]["cond"] == base["cond"]
    assert mem[179]["alloc"] <= base["alloc"] * 1.02 + (16 << 20) and mem[179]["peak"] <= base["peak"] * 1.02 + (16 << 20), (mem, base)
    record("config45_large_default_schedule", frames=n, peak_gib=base["peak"] / 2 ** 30, alloc_gib=base["alloc"] / 2 ** 30,
           tracked=pred.stats["tracked_frames"], encoder_runs=pred.stats["encoder_runs"])
    assert pred.stats["encoder_runs"] == n            # every stream frame encoded once although it is tracked twice
