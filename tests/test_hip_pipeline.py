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


def _iou(a, b):
    inter, union = np.logical_and(a, b).sum(), np.logical_or(a, b).sum()
    return 1.0 if union == 0 else inter / union


@pytest.mark.parametrize("which", ["small", "cut", "default"])
def test_pipeline_matches_reference_pipeline(golden_dir, which):
    """F1 against the REFERENCE's DetSAM2Pipeline.inference (Det_SAM2_pipeline.py:81-247) itself: fixtures pipeline_*.npz were
    recorded by oracle/make_goldens.py `pipeline:<which>` running that class (constructor, producer thread with the
    cv2.VideoCapture loop, transform_video_segments, consumer thread) over the same synthetic stream with a recording
    post-processor.  Compared: the enqueue order of frames_queue, the order of accepted deliveries incl. the corrected second
    visits, has_processed_frames, the special-class boxes handed to get_hole_name, what is left in both dicts, and the mask
    of every object in EVERY delivery (1 - IoU <= 1e-3; pixel counts at full resolution).
    `small` / `cut`: VideoProcessor keyword arguments overridden to buffers of 4 (stream end -> flush, resp. max_frames reached
    with frames still buffered -> no flush); `default`: the reference constructor's own 30 / 30 / 60 / 2000 over 70 frames."""
    import os

    from _util import record
    from det_sam2_amd.Det_SAM2_pipeline import DetSAM2Pipeline
    from oracle.make_goldens import PIPE_IDS
    g = np.load(os.path.join(golden_dir, f"pipeline_{which}.npz"))
    buf, detect, track, keep = (int(x) for x in g["vp_kwargs"])
    n = int(g["n_frames"])
    rec = Recorder()
    kw = {} if which == "default" else dict(frame_buffer_size=buf, detect_interval=detect, max_frame_num_to_track=track,
                                           max_inference_state_frames=keep)
    pipe = DetSAM2Pipeline(sam2_config_path=TINY, detector=SyntheticDetector(len(PIPE_IDS), class_ids=PIPE_IDS), predictor=_pred(),
                           post_processor=rec, **kw)
    vp = pipe.video_processor
    assert (vp.frame_buffer_size, vp.detect_interval, vp.max_frame_num_to_track, vp.max_inference_state_frames) == (buf, detect, track, keep)
    assert vp.skip_classes == {11, 14, 15, 19}
    pipe.inference((synthetic_frame(t) for t in range(n)), max_frames=int(g["max_frames"]), wait=True)
    assert pipe.delivery_log == list(g["enqueued"])
    assert [t for t, _ in rec.deliveries] == list(g["delivered"])
    assert pipe.has_processed_frames == list(g["has_processed"])
    assert len(rec.special) == len(g["special"]) and all(np.array_equal(a.reshape(-1), b) for a, b in zip(rec.special, g["special"]))
    assert sorted(pipe.video_segments) == list(g["left_in_pipeline"]) and sorted(vp.video_segments) == list(g["left_in_backbone"])
    objs = [int(o) for o in g["obj_ids"]]
    worst, worst_area = 0.0, 0
    for i, (t, seg) in enumerate(rec.deliveries):
        assert sorted(seg) == objs
        ref = np.unpackbits(g[f"bits{i}"])[: len(objs) * 512 * 512].reshape(len(objs), 1, 512, 512).astype(bool)
        for j, o in enumerate(objs):
            assert seg[o].shape == (1, 1024, 1024) and seg[o].dtype == bool
            worst = max(worst, 1.0 - _iou(seg[o][:, ::2, ::2], ref[j]))
            worst_area = max(worst_area, abs(int(seg[o].sum()) - int(g[f"area{i}"][j])))
    record("pipeline_ref", which=which, one_minus_iou=worst, max_area_diff_px=worst_area, deliveries=len(rec.deliveries))
    assert worst <= 1e-3, (worst, worst_area)
