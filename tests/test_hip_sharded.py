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
    world = 2
    vps = [P.ShardedVideoProcessor(detector=SyntheticDetector(3, appear=appear), predictor=_pred(), rank=r, world_size=world, **kw)
           for r in range(world)]
    P.drive_lockstep(vps, frames)          # the round generators of both ranks, collectives answered in process
    num_passes = n // buf
    assert sorted(vps[0].owned_passes + vps[1].owned_passes) == list(range(num_passes))
    merged = P.merge_segments([v.video_segments for v in vps], buf, track, num_passes, world, n)
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
    assert differing == 0, (worst, differing)      # same kernels, same inputs, same key order: bit-identical masks
    # each rank tracked only its own passes; the pyramid hand-off means every frame was encoded exactly once
    assert vps[0].predictor.stats["tracked_frames"] + vps[1].predictor.stats["tracked_frames"] == seq.predictor.stats["tracked_frames"]
    assert vps[0].predictor.stats["encoder_runs"] + vps[1].predictor.stats["encoder_runs"] == n == seq.predictor.stats["encoder_runs"]
    ring = sum(b for v in vps for _, op, b in v.comm_log if op == "ring_shift")
    assert ring == n * 16 * 2 ** 20             # every buffer travels once to the owner of the next pass: 16 MiB of pyramids per frame
