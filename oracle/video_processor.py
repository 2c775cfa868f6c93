"""ORACLE (test infrastructure, not product): restatement of the Det-SAM2 streaming driver
``VideoProcessor`` (det_sam2_inference/det_sam2_RT.py:25-651) -- SURVEY.md section 8a rows
A1, A2 -- on top of ``OraclePredictor``.  YOLO is out of scope (third-party); detections are
injected at its output contract ``{"coordinates": xyxy, "class": [c], "confidence": [p]}``
(det_sam2_RT.py:228-244) through ``detector(frame_abs_idx, frame_rgb) -> list[dict]``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg import this.
"""
from __future__ import annotations

import numpy as np

from .predictor import OraclePredictor


class OracleVideoProcessor:
    def __init__(self, sd, cfg, detector, skip_classes=frozenset({11, 14, 15, 19}), frame_buffer_size=30,
                 detect_interval=30, max_frame_num_to_track=60, max_inference_state_frames=60,
                 release_images=True, fill_hole_area=0):
        self.predictor = OraclePredictor(sd, cfg, fill_hole_area=fill_hole_area)
        self.detector = detector
        self.skip_classes = set(skip_classes)
        self.frame_buffer_size, self.detect_interval = frame_buffer_size, detect_interval
        self.max_frame_num_to_track = max_frame_num_to_track
        self.max_inference_state_frames = max_inference_state_frames
        self.release_images = release_images
        self.special_classes = 11
        self.special_classes_detection, self.special_classes_count = [], 0
        self.frame_buffer, self.video_segments, self.inference_state = [], {}, None
        self.pre_frames = 0
        self.pass_log = []  # (start idx, frames yielded, cond keys, non-cond keys) per pass

    def detect_predict(self, images, past_num_frames):
        """det_sam2_RT.py:201-265: detect on frames whose ABSOLUTE index % detect_interval == 0."""
        res = {}
        if self.detect_interval == -1:
            return res
        for i, image in enumerate(images):
            t = past_num_frames + i
            if t % self.detect_interval != 0:
                continue
            dets = list(self.detector(t, image))
            if not self.special_classes_detection:
                self.special_classes_count = 0
            n_special = sum(1 for d in dets if int(np.asarray(d["class"]).reshape(-1)[0]) == self.special_classes)
            if n_special > self.special_classes_count:
                self.special_classes_detection = [d["coordinates"] for d in dets
                                                  if int(np.asarray(d["class"]).reshape(-1)[0]) == self.special_classes]
                self.special_classes_count = n_special
            res[f"frame_{t}"] = dets
        return res

    def detect_2_sam2_prompt(self, detection_results):
        """det_sam2_RT.py:267-316."""
        for key, dets in detection_results.items():
            t = int(key.replace("frame_", ""))
            for d in dets:
                c = int(np.asarray(d["class"]).reshape(-1)[0])
                if c in self.skip_classes:
                    continue
                self.predictor.add_new_points_or_box(self.inference_state, t, c,
                                                     box=np.array(d["coordinates"], dtype=np.float32))

    def detect_and_sam2_inference(self, frame_idx):
        """det_sam2_RT.py:342-411."""
        past = self.inference_state["num_frames"] if self.inference_state else 0
        dets = self.detect_predict(self.frame_buffer, past)
        if self.inference_state is None:
            self.inference_state = self.predictor.init_state(self.frame_buffer)
        else:
            self.inference_state = self.predictor.update_state(self.frame_buffer, self.inference_state)
        self.detect_2_sam2_prompt(dets)
        yielded = []
        for t, obj_ids, logits in self.predictor.propagate_in_video(
                self.inference_state, start_frame_idx=frame_idx,
                max_frame_num_to_track=self.max_frame_num_to_track, reverse=True):
            yielded.append(t)
            if t >= self.pre_frames:
                self.video_segments[t] = {oid: (logits[i] > 0.0).cpu().numpy() for i, oid in enumerate(obj_ids)}
        if self.max_inference_state_frames != -1:
            self.predictor.release_old_frames(self.inference_state, frame_idx, self.max_inference_state_frames,
                                              self.pre_frames, release_images=self.release_images)
        od = self.inference_state["output_dict"]
        self.pass_log.append((frame_idx, yielded, sorted(od["cond_frame_outputs"]), sorted(od["non_cond_frame_outputs"])))

    def process_frame(self, frame_idx, frame):
        """det_sam2_RT.py:421-435."""
        self.frame_buffer.append(frame)
        if len(self.frame_buffer) >= self.frame_buffer_size:
            self.detect_and_sam2_inference(frame_idx)
            self.frame_buffer.clear()
        return self.inference_state

    # ---- A18: preload memory bank (det_sam2_RT.py:489-503 and the prologue of run, :539-549)
    def save_inference_state(self, save_path):
        import pickle
        with open(save_path, "wb") as f:
            pickle.dump(self.inference_state, f)

    def load_inference_state(self, load_path):
        import pickle
        with open(load_path, "rb") as f:
            return pickle.load(f)

    def preload(self, load_path):
        """run() prologue :539-549 (init_preloading_state only moves tensors between devices: nothing to do on CPU)."""
        self.inference_state = self.load_inference_state(load_path)
        od = self.inference_state["output_dict"]
        self.inference_state["preloading_memory_cond_frame_idx"] = list(od["cond_frame_outputs"].keys())
        self.inference_state["preloading_memory_non_cond_frames_idx"] = list(od["non_cond_frame_outputs"].keys())
        self.pre_frames = self.inference_state["num_frames"]

    def run(self, frames):
        """det_sam2_RT.py:526-615 for an in-memory RGB frame list (the cv2.VideoCapture loop :558-579)."""
        idx = 0
        for fr in frames:
            self.process_frame(self.pre_frames + idx, fr)
            idx += 1
        if self.frame_buffer:
            self.detect_and_sam2_inference(self.pre_frames + idx - 1)
            self.frame_buffer.clear()
        self.video_segments = {t - self.pre_frames: s for t, s in self.video_segments.items() if t >= self.pre_frames}
        return self.video_segments
