"""F1: DetSAM2Pipeline hand-off contract (Det_SAM2_pipeline.py:59-78,183-214 of the reference) with a stub backbone -
CPU only.  The GPU test (tests/test_hip_pipeline.py) runs the real VideoProcessor behind it."""
import numpy as np
import pytest

from det_sam2_amd.Det_SAM2_pipeline import DetSAM2Pipeline
from det_sam2_amd.det_sam2_RT import PackedMasks


class StubBackbone:
    """Mimics VideoProcessor's observable behaviour: buffers `buf` frames, then a reverse pass over the last 2*buf
    frames rewrites video_segments for all of them (det_sam2_RT.py:388-399)."""

    def __init__(self, buf=3, special_at=0, pre_frames=0):
        self.buf, self.special_at, self.pre_frames = buf, special_at, pre_frames
        self.frame_buffer, self.video_segments, self.special_classes_detection = [], {}, []
        self.load_inference_state_path, self.vis_frame_stride, self.inference_state = None, -1, None
        self.passes = 0

    def _pass(self, frame_idx):
        self.passes += 1
        for t in range(max(self.pre_frames, frame_idx - 2 * self.buf + 1), frame_idx + 1):
            bits = np.full((2, 4, 1), self.passes, np.uint8)                    # the pass number is the payload
            self.video_segments[t] = PackedMasks(bits, [5, 9], 8)
        if frame_idx >= self.special_at:
            self.special_classes_detection = [np.zeros(4, np.float32)]

    def process_frame(self, frame_idx, frame):
        self.frame_buffer.append(frame)
        if len(self.frame_buffer) >= self.buf:
            self._pass(frame_idx)
            self.frame_buffer.clear()
        return self.inference_state

    def Detect_and_SAM2_inference(self, frame_idx):
        self._pass(frame_idx)


class Recorder:
    def __init__(self):
        self.started_with, self.seen = None, []

    def start(self, special):
        self.started_with = list(special)

    def process(self, frame_idx, segments):
        assert sorted(segments) == [5, 9] and segments[5].shape == (1, 4, 8) and segments[5].dtype == bool
        self.seen.append((frame_idx, int(segments.packed[0, 0, 0])))


def test_each_frame_is_delivered_twice_in_order():
    vp, rec = StubBackbone(buf=3), Recorder()
    pipe = DetSAM2Pipeline(video_processor=vp, post_processor=rec)
    pipe.inference(iter([None] * 11), max_frames=100, wait=True)           # 3 full passes + a flush of 2 buffered frames
    assert vp.passes == 4
    # enqueue order: every pass delivers its whole window in ascending absolute index (transform_video_segments :65-78)
    assert pipe.delivery_log == [0, 1, 2] + [0, 1, 2, 3, 4, 5] + [3, 4, 5, 6, 7, 8] + [5, 6, 7, 8, 9, 10]
    # consumer: first visit in order, then the corrected re-delivery; nothing skipped ahead (:191)
    assert pipe.has_processed_frames == list(range(11))
    first = {}
    for t, p in rec.seen:
        first.setdefault(t, p)
    assert all(first[t] <= p for t, p in rec.seen)
    assert [t for t, _ in rec.seen] == pipe.delivery_log
    assert rec.started_with is not None and pipe.video_segments == {} and vp.video_segments == {}


def test_consumer_starts_with_the_first_special_detection_and_catches_up():
    vp, rec = StubBackbone(buf=2, special_at=5), Recorder()
    pipe = DetSAM2Pipeline(video_processor=vp, post_processor=rec)
    pipe.inference(iter([None] * 8), max_frames=100, wait=True)
    assert pipe.post_processor_started and pipe.has_processed_frames == list(range(8))
    assert [t for t, _ in rec.seen] == pipe.delivery_log                   # the queue kept everything delivered before the start


def test_without_special_detection_nothing_is_consumed_like_the_reference():
    vp, rec = StubBackbone(buf=2, special_at=10 ** 9), Recorder()
    pipe = DetSAM2Pipeline(video_processor=vp, post_processor=rec)
    pipe.inference(iter([None] * 4), max_frames=100, wait=True)
    assert not pipe.post_processor_started and rec.seen == [] and pipe.frames_queue.qsize() == len(pipe.delivery_log)


def test_max_frames_and_preload_offset():
    vp, rec = StubBackbone(buf=2, pre_frames=7), Recorder()
    pipe = DetSAM2Pipeline(video_processor=vp, post_processor=rec, start_postprocess="immediately")
    pipe.inference(iter([None] * 100), max_frames=6, wait=True)             # stops after 6 stream frames (:157-159)
    assert vp.passes == 3 and pipe.has_processed_frames == list(range(6))  # consumer sees stream-relative indices (:188)
    assert min(pipe.delivery_log) == 7


@pytest.mark.parametrize("which", ["small", "cut", "default"])
def test_delivery_order_matches_the_reference_pipeline(golden_dir, which):
    """The hand-off's ORDER against what the reference's own DetSAM2Pipeline.inference did (fixtures pipeline_*.npz recorded
    by oracle/make_goldens.py `pipeline:<which>` from /root/reference/det_sam2_inference/Det_SAM2_pipeline.py): the queue's
    enqueue order, the consumer's accepted deliveries and has_processed_frames.  The backbone here is the stub with the
    fixture's buffer size; the masks themselves are compared on the GPU (tests/test_hip_pipeline.py)."""
    import os
    g = np.load(os.path.join(golden_dir, f"pipeline_{which}.npz"))
    buf, detect, track, _ = (int(x) for x in g["vp_kwargs"])
    assert track == 2 * buf and detect == buf
    vp, rec = StubBackbone(buf=buf), Recorder()
    pipe = DetSAM2Pipeline(video_processor=vp, post_processor=rec)
    pipe.inference(iter([None] * int(g["n_frames"])), max_frames=int(g["max_frames"]), wait=True)
    assert pipe.delivery_log == list(g["enqueued"])
    assert [t for t, _ in rec.seen] == list(g["delivered"])
    assert pipe.has_processed_frames == list(g["has_processed"])
    assert sorted(pipe.video_segments) == list(g["left_in_pipeline"]) and sorted(vp.video_segments) == list(g["left_in_backbone"])
