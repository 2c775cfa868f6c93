"""F1 on the GPU: DetSAM2Pipeline over the real VideoProcessor (tiny model).  The consumer receives every stream frame
twice - first tracking, then the corrected second visit (Det_SAM2_pipeline.py:190-191) - in the reference's wire format
(frame_idx, {obj_id: bool[1,Hv,Wv]}); the LAST delivery of a frame equals what the plain streaming driver ends with."""
import numpy as np
import pytest

from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
TINY = "sam2.1_hiera_t"
KW = dict(frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=8, max_inference_state_frames=8)


def _pred():
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config(TINY)
    return SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=4)


class Recorder:
    def __init__(self):
        self.deliveries, self.special = [], None

    def start(self, special):
        self.special = [np.asarray(s) for s in special]

    def process(self, frame_idx, segments):
        self.deliveries.append((frame_idx, {oid: segments[oid].copy() for oid in segments}))


def test_pipeline_redelivers_corrected_frames():
    from det_sam2_amd.Det_SAM2_pipeline import DetSAM2Pipeline
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    n = 12
    det = lambda: SyntheticDetector(3, class_ids=[2, 11, 6])          # class 11: special (collected, skipped by default)  # noqa: E731
    frames = [synthetic_frame(t) for t in range(n)]
    seq = VideoProcessor(model_cfg=TINY, detector=det(), predictor=_pred(), **KW)
    for t, f in enumerate(frames):
        seq.process_frame(t, f)
    rec = Recorder()
    pipe = DetSAM2Pipeline(sam2_config_path=TINY, detector=det(), predictor=_pred(), post_processor=rec, **KW)
    pipe.inference(iter(frames), max_frames=1000, wait=True)
    assert rec.special is not None and len(rec.special) == 1                 # post-processing started on the class-11 box
    assert pipe.has_processed_frames == list(range(n))
    count, last = {}, {}
    for t, seg in rec.deliveries:
        assert sorted(seg) == [2, 6] and all(m.shape == (1, 1024, 1024) and m.dtype == bool for m in seg.values())
        count[t] = count.get(t, 0) + 1
        last[t] = seg
    assert [count[t] for t in range(n)] == [2] * (n - 4) + [1] * 4          # the newest buffer has only been tracked once
    for t in range(n):
        for oid in (2, 6):
            assert np.array_equal(last[t][oid], seq.video_segments[t][oid]), (t, oid)
    assert pipe.video_segments == {}                                        # consumed frames are dropped (:211-213)
