"""The pass-sharded streaming driver (det_sam2_amd.parallel) on CPU: the real predictor state machine and the
real round logic over a CPU stand-in for the HIP stages (tests/_fake_hip.py).  In-process lock-step ranks here; the same
rounds through torch.distributed (gloo, 2 processes) in tests/test_parallel_gloo.py."""
import numpy as np
import pytest

from _fake_hip import fake_predictor
from det_sam2_amd import parallel as P
from det_sam2_amd.det_sam2_RT import VideoProcessor
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame

KW = dict(model_cfg="sam2.1_hiera_t", frame_buffer_size=3, detect_interval=3, max_frame_num_to_track=6, max_inference_state_frames=6)


def _det(appear=None):
    return SyntheticDetector(4, size=32, appear=appear, class_ids=[5, 11, 2, 8])     # class 11 = special, skipped


def _sequential(frames, appear, **over):
    vp = VideoProcessor(detector=_det(appear), predictor=fake_predictor(), **{**KW, **over})
    vp.run(frames=frames)
    return vp


def _sharded(frames, appear, world, handoff=True, **over):
    vps = [P.ShardedVideoProcessor(detector=_det(appear), predictor=fake_predictor(), rank=r, world_size=world,
                                   handoff_features=handoff, **{**KW, **over}) for r in range(world)]
    P.drive_lockstep(vps, frames)
    return vps


def _same(a, b):
    assert sorted(a) == sorted(b)
    for t in a:
        assert sorted(a[t]) == sorted(b[t]), t
        for o in a[t]:
            assert np.array_equal(a[t][o], b[t][o]), (t, o)


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("appear", [None, {3: 7}])
@pytest.mark.parametrize("n", [18, 20])                 # 20: the stream ends with a partial buffer (and a partial round)
def test_sharded_rounds_equal_sequential(world, appear, n):
    frames = [synthetic_frame(t, size=32) for t in range(n)]
    seq = _sequential(frames, appear)
    vps = _sharded(frames, appear, world)
    num_passes = -(-n // 3)
    assert sorted(k for v in vps for k in v.owned_passes) == list(range(num_passes))
    for v in vps:
        assert v.owned_passes == P.passes_of_rank(num_passes, world, v.rank)
    merged = P.merge_segments([v.video_segments for v in vps], 3, 6, num_passes, world, n)
    _same(merged, seq.video_segments)
    # a frame is encoded exactly once per stream (pyramid hand-off), every tracked frame is tracked exactly once
    enc = sum(v.predictor.stats["encoder_runs"] for v in vps)
    assert seq.predictor.stats["encoder_runs"] == n
    # (the window of a final PARTIAL buffer reaches one buffer further back than the hand-off covers)
    assert enc == n if n % 3 == 0 else n <= enc <= n + 2, enc
    assert sum(v.predictor.stats["tracked_frames"] for v in vps) == seq.predictor.stats["tracked_frames"]
    # replicated bookkeeping: same object table, same live bank keys, same retained frames, same special-class state
    for v in vps:
        assert v.inference_state["obj_ids"] == seq.inference_state["obj_ids"]
        assert sorted(v.inference_state["output_dict"]["cond_frame_outputs"]) == sorted(seq.inference_state["output_dict"]["cond_frame_outputs"])
        assert set(v.inference_state["images_idx"]) <= set(seq.inference_state["images_idx"])   # a rank ingests only its own buffers
        assert v.special_classes_count == seq.special_classes_count
        assert np.array_equal(np.asarray(v.special_classes_detection), np.asarray(seq.special_classes_detection))
        assert [p[0] for p in v.pass_log] == [p[0] for p in seq.pass_log]
        for t, e in v.inference_state["output_dict"]["cond_frame_outputs"].items():
            ref = seq.inference_state["output_dict"]["cond_frame_outputs"][t]
            for k in P.ENTRY_FIELDS:
                assert e[k].shape == ref[k].shape and bool((e[k] == ref[k]).all()), (t, k)


def test_without_handoff_every_frame_is_encoded_by_both_passes():
    n = 12
    frames = [synthetic_frame(t, size=32) for t in range(n)]
    seq = _sequential(frames, None)
    vps = _sharded(frames, None, 2, handoff=False)
    _same(P.merge_segments([v.video_segments for v in vps], 3, 6, 4, 2), seq.video_segments)
    assert sum(v.predictor.stats["encoder_runs"] for v in vps) > n


def test_world_size_one_is_the_sequential_driver():
    frames = [synthetic_frame(t, size=32) for t in range(10)]
    seq = _sequential(frames, {3: 4})
    one = P.ShardedVideoProcessor(detector=_det({3: 4}), predictor=fake_predictor(), rank=0, world_size=1, **KW)
    one.run(frames=frames)
    _same(one.video_segments, seq.video_segments)


def test_no_release_and_detect_every_frame():
    """max_inference_state_frames = -1 (bank-building mode) and several conditioning frames per pass."""
    frames = [synthetic_frame(t, size=32) for t in range(9)]
    over = dict(detect_interval=1, max_inference_state_frames=-1)
    seq = _sequential(frames, {2: 4}, **over)
    vps = _sharded(frames, {2: 4}, 2, **over)
    _same(P.merge_segments([v.video_segments for v in vps], 3, 6, 3, 2), seq.video_segments)


def test_more_than_255_detections_in_one_pass():
    """the detection all-gather had a fixed 255-row wire format; a pass with detect_interval 1 and many boxes per
    frame overflowed it while the sequential driver handled the same stream.  The row count is now agreed on per round."""
    det = lambda: SyntheticDetector(3, size=32, duplicates={0: 20, 1: 20, 2: 20}, class_ids=[5, 2, 8])   # noqa: E731  63 boxes / frame
    over = dict(frame_buffer_size=5, detect_interval=1, max_frame_num_to_track=10, max_inference_state_frames=-1)
    frames = [synthetic_frame(t, size=32) for t in range(10)]
    assert sum(len(det()(t)) for t in range(5)) == 315
    seq = VideoProcessor(detector=det(), predictor=fake_predictor(), **{**KW, **over})
    seq.run(frames=frames)
    vps = [P.ShardedVideoProcessor(detector=det(), predictor=fake_predictor(), rank=r, world_size=2, **{**KW, **over}) for r in range(2)]
    P.drive_lockstep(vps, frames)
    _same(P.merge_segments([v.video_segments for v in vps], 5, 10, 2, 2), seq.video_segments)
    assert any(kind == "all_gather_dets" and nbytes > 2 * 256 * 7 * 8 for v in vps for _, kind, nbytes in v.comm_log)
