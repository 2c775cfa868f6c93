"""BASELINE configs 3, 4 and 5 at their REAL model / object-count sizes on the GPU (the oracle would need hours here, so
these are the size-independent properties; numerics at these sizes are pinned by the reference goldens e2e_large,
e2e_b16, e2e_b17 and the module fixtures of all four configs):

* config 3: sam2.1_hiera_base_plus, 16 objects, a PRELOADED bank of P conditioning frames (P = 1: the "7-frame bank";
  P = 10: the reference's example size, det_sam2_RT.py:660-665) written and read as a DS2BANK file, no detector
  afterwards (detect_interval = -1): every tracked frame attends Nk = 4096 (P + 6) + 64 keys in steady state;
* config 4 / 5: sam2.1_hiera_large, 16 objects, Det-SAM2's default 30/30/60/60 schedule over 180 frames with a 17th
  class first detected at stream frame 90 (A17 at full batch: workspace growth past max_batch = 16); the retained state
  and the HBM footprint are flat once the window is full ("assert peak VRAM flat after frame 120")."""
import time

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
    n = 500                                  # BASELINE config 3's stated stream length
    b = VideoProcessor(model_cfg=cfg.name, detector=SyntheticDetector(16), skip_classes=set(), predictor=pred, frame_buffer_size=15,
                       detect_interval=-1, max_frame_num_to_track=30, max_inference_state_frames=30, load_inference_state_path=bank)
    pred.trace = []
    enc0, trk0 = pred.stats["encoder_runs"], pred.stats["tracked_frames"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    segs = b.run(frames=(synthetic_frame(200 + t) for t in range(n)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    trace, pred.trace = pred.trace, None
    assert pred.stats["encoder_runs"] - enc0 == n       # every stream frame encoded exactly once although it is tracked twice
    # no conditioning frames in the new stream: every visited frame is tracked; pass k visits min(30, frames so far)
    assert pred.stats["tracked_frames"] - trk0 == sum(min(30, min(15 * (k + 1), n)) for k in range(-(-n // 15)))
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
    # constant state: the window never holds more than 2 buffers + the bank
    st = b.inference_state
    assert len(st["output_dict"]["non_cond_frame_outputs"]) <= 45 and len(st["output_dict"]["cond_frame_outputs"]) == P
    record("config3_bplus_preload", P=P, nk=nk_full, frames=n, tracked=len(trace), seconds=dt, stream_fps=n / dt, tracked_fps=len(trace) / dt)


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


def test_config4_large_16_objects_1000_frames_sharded_over_8_ranks_equals_sequential():
    """BASELINE config 4 at its stated size: sam2.1_hiera_large, ONE 1000-frame stream, 16 objects, the default
    30/30/60/60 schedule, propagate passes sharded over 8 ranks (det_sam2_amd.parallel.ShardedVideoProcessor; pass k -> rank
    k mod 8).  A single MI355X holds all 8 predictors (288 GB), their rounds run in lock step with the collectives answered
    in process (run_lockstep: same round generator as under RCCL, tests/test_parallel_gloo.py runs it through
    torch.distributed).  Asserted: masks BIT-IDENTICAL to the sequential driver on every 7th frame, every frame encoded once
    per stream (pyramid hand-off), device memory flat from round to round; the bytes each collective moves are recorded."""
    from det_sam2_amd import parallel as P
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config("sam2.1_hiera_large")
    sd = synthetic_state_dict(cfg, 0)
    n, world, buf, B = 1000, 8, 30, 16
    sample = set(range(0, n, 7)) | {n - 1}

    def frames():
        return (synthetic_frame(t) for t in range(n))

    def keep_sample(segs):
        for t in [t for t in segs if t not in sample]:
            del segs[t]

    # ---- sequential reference (async encoder as shipped)
    pred = SAM2VideoPredictor(cfg, sd, "cuda:0", max_batch=B)
    seq = VideoProcessor(model_cfg=cfg.name, detector=SyntheticDetector(B), skip_classes=set(), predictor=pred)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t, f in enumerate(frames()):
        seq.process_frame(t, f)
        if t % 120 == 119:
            keep_sample(seq.video_segments)
    if seq.frame_buffer:
        seq.Detect_and_SAM2_inference(frame_idx=n - 1)
        seq.frame_buffer.clear()
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    keep_sample(seq.video_segments)
    ref = {t: {o: np.packbits(m, axis=-1) for o, m in seq.video_segments[t].items()} for t in sorted(seq.video_segments)}
    enc_seq, trk_seq = pred.stats["encoder_runs"], pred.stats["tracked_frames"]
    assert enc_seq == n
    del seq, pred
    torch.cuda.empty_cache()
    # ---- 8 ranks in lock step
    vps = []
    for r in range(world):
        p = SAM2VideoPredictor(cfg, sd, "cuda:0", max_batch=B)
        p.async_encode = False                 # a sharded rank encodes its whole buffer up front (encode_frames)
        p.encode_batch = 5                     # 8 predictors share ONE GPU here: keep each encoder arena at ~10 GB (same results)
        vps.append(P.ShardedVideoProcessor(model_cfg=cfg.name, detector=SyntheticDetector(B), skip_classes=set(), predictor=p,
                                           rank=r, world_size=world))
    free = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = -1
    for idx, fr in enumerate(frames()):
        for v in vps:
            v.frame_buffer.append(fr)
        if len(vps[0].frame_buffer) >= buf * world:
            P.run_lockstep([v.round_generator(idx) for v in vps])
            for v in vps:
                v.frame_buffer.clear()
                keep_sample(v.video_segments)
            torch.cuda.synchronize()
            free.append(torch.cuda.mem_get_info()[0])
    if vps[0].frame_buffer:
        P.run_lockstep([v.round_generator(idx) for v in vps])
        for v in vps:
            v.frame_buffer.clear()
            keep_sample(v.video_segments)
    torch.cuda.synchronize()
    t_sh = time.perf_counter() - t0
    num_passes = -(-n // buf)
    assert sorted(k for v in vps for k in v.owned_passes) == list(range(num_passes))
    merged = P.merge_segments([v.video_segments for v in vps], buf, 2 * buf, num_passes, world, n)
    assert sorted(merged) == sorted(ref)
    differing = 0
    for t in sorted(ref):
        assert sorted(merged[t]) == sorted(ref[t]) == list(range(B)), t
        for o in ref[t]:
            differing += int((np.packbits(merged[t][o], axis=-1) != ref[t][o]).sum())
    enc = sum(v.predictor.stats["encoder_runs"] for v in vps)
    trk = sum(v.predictor.stats["tracked_frames"] for v in vps)
    comm = {}
    for _, op, nbytes in vps[0].comm_log:
        comm[op] = max(comm.get(op, 0), nbytes)
    record("config4_large_1000_frames_world8", frames=n, sampled_frames=len(ref), differing_bytes=differing, encoder_runs=enc,
           tracked=trk, seconds_sequential=t_seq, seconds_sharded_lockstep_one_gpu=t_sh, free_gib_after_rounds=[f / 2 ** 30 for f in free],
           comm_bytes_per_round=comm)
    assert differing == 0                                  # same kernels, same inputs, same key order: bit-identical
    assert trk == trk_seq
    assert n <= enc <= n + buf, enc                        # (the window of the final PARTIAL buffer reaches one buffer further back)
    # flat HBM: free memory after rounds 2.. stays within 1 GiB of the level after round 1 (round 0 grows the arenas)
    assert len(free) == 4 and all(abs(f - free[1]) < (1 << 30) for f in free[2:]), free
    assert comm["ring_shift"] == buf * 16 * 2 ** 20        # a buffer of pyramids travels once per rank and round
