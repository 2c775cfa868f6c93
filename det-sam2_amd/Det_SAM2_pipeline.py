"""DetSAM2Pipeline: the streaming hand-off between the inference backbone and its consumer
(det_sam2_inference/Det_SAM2_pipeline.py:18-247).

What is kept from the reference, because downstream code depends on it:

* the wire format - ``frames_queue`` carries ``(absolute_frame_idx, {obj_id: bool[1,Hv,Wv]})`` (:59-78); here the
  mapping is a ``PackedMasks`` view over the bit-packed rows that left the GPU (same reads, 8x smaller);
* ``transform_video_segments`` moves everything ``VideoProcessor`` produced since the last call into the pipeline's
  own dict, clears the producer's dict, and enqueues the frames in ascending index order (:65-78) - so every stream
  frame is DELIVERED TWICE: once by the pass that first tracks it and again, corrected, by the next pass whose
  60-frame reverse window covers it (max_frame_num_to_track = 2 x frame_buffer_size, :190-191);
* the consumer accepts a frame only if ``frame_idx <= len(has_processed_frames)``: re-deliveries of already processed
  frames are processed again, frames are never skipped ahead (:191); it starts when the first special-class
  detection exists (:147-155) and ends when inference is done and the queue is empty (:183-185).

What is not rebuilt: cv2 video decoding (frames are handed in as an iterable), rendering, and the billiards event logic
(``VideoPostProcessor``: out of scope) - the consumer is any object with ``process(frame_idx, segments)``
(and optionally ``start(special_classes_detection)``); an object exposing the reference post-processor's method names
is driven the way the reference drives it.
"""
from __future__ import annotations

import threading
from queue import Empty, Queue

from .det_sam2_RT import VideoProcessor


class _ReferencePostProcessorAdapter:
    """Drives an object with the reference VideoPostProcessor's method names (Det_SAM2_pipeline.py:150-151,193-208)."""

    def __init__(self, pp):
        self.pp = pp

    def start(self, special):
        self.pp.get_hole_name(special)
        self.pp.get_boundary_from_holes()

    def process(self, frame_idx, segments):
        pp = self.pp
        pp.balls_positions[frame_idx] = pp.process_frame_positions(segments)
        if frame_idx > 0:
            pp.balls_velocities[frame_idx] = pp.process_frame_velocities(frame_idx, time_interval=1.0)
            pp.check_ball_disappeared_pot(frame_idx)
            if frame_idx > 1:
                pp.check_ball_collision(frame_idx)
                pp.check_ball_rebound(frame_idx)


class DetSAM2Pipeline:
    def __init__(self, sam2_output_frame_dir=None, sam2_checkpoint_path=None,
                 sam2_config_path="configs/sam2.1/sam2.1_hiera_l.yaml", detect_model_weights=None, output_video_dir=None,
                 load_inference_state_path=None, visualize_postprocessor=False, *, post_processor=None, detector=None,
                 predictor=None, video_processor=None, device="cuda", start_postprocess="special", **video_processor_kwargs):
        if visualize_postprocessor:
            raise NotImplementedError("post-processing visualisation is outside the hot path")
        if video_processor is None:
            kw = dict(skip_classes={11, 14, 15, 19}, vis_frame_stride=-1, visualize_prompt=False, frame_buffer_size=30,
                      detect_interval=30, max_frame_num_to_track=60, max_inference_state_frames=2000)   # :39-46
            kw.update(video_processor_kwargs)
            video_processor = VideoProcessor(output_dir=sam2_output_frame_dir, sam2_checkpoint=sam2_checkpoint_path,
                                             model_cfg=sam2_config_path, detect_model_weights=detect_model_weights,
                                             load_inference_state_path=load_inference_state_path, detector=detector,
                                             predictor=predictor, device=device, **kw)
        self.video_processor = video_processor
        if post_processor is not None and not hasattr(post_processor, "process") and hasattr(post_processor, "process_frame_positions"):
            post_processor = _ReferencePostProcessorAdapter(post_processor)
        self.post_processor = post_processor
        self.output_video_dir = output_video_dir
        assert start_postprocess in ("special", "immediately")
        self.start_postprocess = start_postprocess
        self.inference_done_event = threading.Event()
        self.video_segments = {}            # the pipeline's own copy: the two threads never share VideoProcessor's dict
        self.frames_queue = Queue()
        self.has_processed_frames = []
        self.delivery_log = []              # absolute frame indices in the order they were enqueued
        self._lock = threading.Lock()       # the reference's readerwriterlock write lock (:71)
        self.post_processor_started = False
        self._threads = []
        self.error = None

    # ------------------------------------------------------------------ hand-off (:59-78)
    def transform_video_segments(self):
        vp = self.video_processor
        with self._lock:
            # snapshot under the lock: the consumer pops delivered frames from self.video_segments concurrently, and an
            # OLD delivery of frame t still draining would otherwise remove the entry before it is enqueued here
            items = [(t, vp.video_segments[t]) for t in sorted(vp.video_segments)]
            self.video_segments.update(vp.video_segments)
            vp.video_segments.clear()
        for t, seg in items:
            self.frames_queue.put((t, seg))
            self.delivery_log.append(t)

    # ------------------------------------------------------------------ the two threads (:81-247)
    def _preload(self):
        vp = self.video_processor
        if vp.load_inference_state_path is None:
            return
        vp.inference_state = vp.load_inference_state(vp.load_inference_state_path)
        od = vp.inference_state["output_dict"]
        vp.inference_state["preloading_memory_cond_frame_idx"] = list(od["cond_frame_outputs"].keys())
        vp.inference_state["preloading_memory_non_cond_frames_idx"] = list(od["non_cond_frame_outputs"].keys())
        vp.pre_frames = vp.inference_state["num_frames"]
        vp.predictor.init_preloading_state(vp.inference_state)

    def _maybe_start_consumer(self, thread):
        if self.post_processor_started:
            return
        vp = self.video_processor
        if self.start_postprocess == "immediately" or vp.special_classes_detection:
            if self.post_processor is not None and hasattr(self.post_processor, "start"):
                self.post_processor.start(vp.special_classes_detection)
            thread.start()
            self.post_processor_started = True

    def _process_video(self, frames, max_frames, consumer_thread):
        vp = self.video_processor
        try:
            self._preload()
            frame_idx = 0
            it = iter(frames)
            while frame_idx < max_frames:
                try:
                    frame_rgb = next(it)
                except StopIteration:          # stream ended: flush what is buffered (:124-131)
                    if vp.frame_buffer:
                        vp.Detect_and_SAM2_inference(frame_idx=vp.pre_frames + frame_idx - 1)
                        vp.frame_buffer.clear()
                        self.transform_video_segments()
                    break
                vp.inference_state = vp.process_frame(vp.pre_frames + frame_idx, frame_rgb)
                self.transform_video_segments()
                self._maybe_start_consumer(consumer_thread)
                frame_idx += 1
        except BaseException as e:             # surface producer failures to join()
            self.error = e
        finally:
            self.inference_done_event.set()

    def _post_process(self):
        vp = self.video_processor
        try:
            while True:
                if self.inference_done_event.is_set() and self.frames_queue.empty():
                    break
                try:
                    frame_idx, segments = self.frames_queue.get(timeout=0.02)
                except Empty:
                    continue
                frame_idx -= vp.pre_frames
                if frame_idx <= len(self.has_processed_frames):      # re-deliveries yes, skipping ahead no (:191)
                    if self.post_processor is not None:
                        self.post_processor.process(frame_idx, segments)
                    if frame_idx not in self.has_processed_frames:
                        self.has_processed_frames.append(frame_idx)
                    if vp.vis_frame_stride == -1:
                        with self._lock:            # keys are ABSOLUTE indices (the reference pops the relative one, :204)
                            self.video_segments.pop(frame_idx + vp.pre_frames, None)
        except BaseException as e:
            self.error = self.error or e

    def inference(self, video_source, max_frames, wait=False):
        """video_source: iterable of HxWx3 uint8 RGB frames.  Starts the producer thread (the consumer thread starts with
        the first special-class detection, as in the reference) and returns; ``wait=True`` (or ``join()``) blocks until
        both are finished."""
        consumer = threading.Thread(target=self._post_process, daemon=True)
        producer = threading.Thread(target=self._process_video, args=(video_source, max_frames, consumer), daemon=True)
        self._threads = [producer, consumer]
        producer.start()
        if wait:
            self.join()

    def join(self):
        producer, consumer = self._threads
        producer.join()
        if consumer.is_alive() or self.post_processor_started:
            consumer.join()
        if self.error is not None:
            raise self.error
