"""SURVEY 8e on the GPU: a stream whose passes are sharded over 2 ranks (emulated in one process, in-memory exchange
of the conditioning-frame entries) produces the SAME masks as the sequential VideoProcessor - including a new object
appearing mid-stream (A17) and frame eviction."""
import numpy as np
import pytest

from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

from _util import record

pytestmark = pytest.mark.gpu
TINY = "sam2.1_hiera_t"


def _pred():
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config(TINY)
    return SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=4)


@pytest.mark.parametrize("appear", [None, {2: 20}])
def test_pass_sharded_stream_equals_sequential(appear):
    from det_sam2_amd import parallel as P
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    buf, track, keep, n = 10, 20, 20, 60
    kw = dict(model_cfg=TINY, skip_classes=set(), frame_buffer_size=buf, detect_interval=buf, max_frame_num_to_track=track,
              max_inference_state_frames=keep)
    frames = [synthetic_frame(t) for t in range(n)]
    seq = VideoProcessor(detector=SyntheticDetector(3, appear=appear), predictor=_pred(), **kw)
    for t, f in enumerate(frames):
        seq.process_frame(t, f)

    box = {}

    def exchange(k, owner, payload):
        if payload is not None:
            box[k] = payload
        return box[k]

    world = 2
    vps = [P.ShardedVideoProcessor(detector=SyntheticDetector(3, appear=appear), predictor=_pred(), rank=r, world_size=world,
                                   exchange=exchange, **kw) for r in range(world)]
    for t, f in enumerate(frames):
        owner = P.pass_owner(t // buf, world)
        for r in [owner] + [x for x in range(world) if x != owner]:     # the owner's flush fills the mailbox first
            vps[r].process_frame(t, f)
    num_passes = n // buf
    assert sorted(vps[0].owned_passes + vps[1].owned_passes) == list(range(num_passes))
    merged = P.merge_segments([v.video_segments for v in vps], buf, track, num_passes, world)
    assert sorted(merged) == sorted(seq.video_segments) == list(range(n))
    worst, differing = 0.0, 0
    for t in range(n):
        assert sorted(merged[t]) == sorted(seq.video_segments[t]), t
        for o in merged[t]:
            a, b = merged[t][o], seq.video_segments[t][o]
            differing += int((a != b).sum())
            u = (a | b).sum()
            worst = max(worst, 1.0 - ((a & b).sum() / u if u else 1.0))
    record("sharded_vs_sequential", appear=str(appear), one_minus_iou=worst, differing_pixels=differing)
    assert differing == 0, (worst, differing)      # same kernels, same inputs: bit-identical masks
    # each rank ran only its own passes' encoders / trackers
    assert vps[0].predictor.stats["tracked_frames"] + vps[1].predictor.stats["tracked_frames"] == seq.predictor.stats["tracked_frames"]
