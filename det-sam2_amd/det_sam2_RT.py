"""VideoProcessor: the Det-SAM2 streaming driver (det_sam2_inference/det_sam2_RT.py:25-651) on MI355X.

Same constructor arguments and public methods (``run``, ``process_frame``,
``Detect_and_SAM2_inference``, ``clear``, ``save/load_inference_state``).  Differences, all outside the
arithmetic: YOLO is injected (``detector``: any callable at the YOLO output contract
det_sam2_RT.py:228-244, or a path handled by ultralytics if it is installed); thresholded masks leave the
GPU bit-packed, once per pass, instead of one blocking ``.cpu()`` per object per frame (:396-399);
visualisation/rendering (:628-651) is not part of the hot path.
"""
from __future__ import annotations

import os
import pickle
from collections.abc import Mapping, Sequence

import numpy as np
import torch

from . import bank_io
from .build_sam import build_sam2_video_predictor


class PackedMasks(Mapping):
    """``{obj_id: bool[1,Hv,Wv]}`` of one frame (the reference's ``video_segments[frame]`` value, det_sam2_RT.py:396-399)
    held as the bit-packed rows that left the GPU (``numpy.packbits`` semantics, 8 px/byte): an object's mask is
    unpacked when it is read, so a 16-object 1080p frame costs 4 MiB on the host instead of 32 MiB and the stream loop
    never unpacks masks nobody looks at.  Behaves like the reference's dict for readers (``[]``, ``in``, iteration,
    ``items()``); ``to_dict()`` gives the plain dict (used for the pickled output, :612-615)."""

    __slots__ = ("packed", "obj_ids", "width", "_pos")

    def __init__(self, packed, obj_ids, width):
        self.packed, self.obj_ids, self.width = packed, list(obj_ids), int(width)   # packed: uint8 [B,Hv,ceil(Wv/8)]
        self._pos = {oid: i for i, oid in enumerate(self.obj_ids)}

    def __getitem__(self, oid):
        row = self.packed[self._pos[oid]]
        return np.unpackbits(row, axis=-1)[:, : self.width].astype(bool)[None]

    def __iter__(self):
        return iter(self.obj_ids)

    def __len__(self):
        return len(self.obj_ids)

    def to_dict(self):
        return {oid: self[oid] for oid in self.obj_ids}


class _FolderFrames(Sequence):
    """Lazy list of the image files of a folder (see VideoProcessor.load_frames_from_folder).  Construction only reads the
    headers (PIL's verify: no pixel is decoded); a frame is decoded when it is accessed.  A file whose header parses but whose
    payload does not decode (a truncated JPEG) is skipped, with the reference's message, when the stream reaches it -
    cv2.imread returning None has the same effect there (det_sam2_RT.py:516-518)."""

    def __init__(self, folder_path):
        from PIL import Image
        self.folder, self.names = folder_path, []
        for name in sorted(f for f in os.listdir(folder_path) if f.endswith((".png", ".jpg", ".jpeg"))):
            try:
                with Image.open(os.path.join(folder_path, name)) as im:
                    im.verify()
                self.names.append(name)
            except Exception:
                print(f"--- cannot read frame file: {os.path.join(folder_path, name)}")

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        from PIL import Image, ImageOps
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        with Image.open(os.path.join(self.folder, self.names[i])) as im:
            return np.ascontiguousarray(np.asarray(ImageOps.exif_transpose(im).convert("RGB"), dtype=np.uint8))

    def __iter__(self):
        for i in range(len(self.names)):
            try:
                yield self[i]
            except Exception:
                print(f"--- cannot read frame file: {os.path.join(self.folder, self.names[i])}")


class VideoProcessor:
    def __init__(self, output_dir=None, sam2_checkpoint=None, model_cfg="configs/sam2.1/sam2.1_hiera_l.yaml",
                 detect_model_weights=None, detect_confidence=0.85, skip_classes={11, 14, 15, 19}, vis_frame_stride=-1,
                 visualize_prompt=False, frame_buffer_size=30, detect_interval=30, max_frame_num_to_track=60,
                 max_inference_state_frames=60, load_inference_state_path=None, save_inference_state_path=None,
                 detector=None, device="cuda", predictor=None):
        self.output_dir = output_dir
        self.sam2_checkpoint, self.model_cfg = sam2_checkpoint, model_cfg
        self.detect_model_weights, self.detect_confidence = detect_model_weights, detect_confidence
        self.skip_classes = set(skip_classes)
        self.vis_frame_stride, self.visualize_prompt = vis_frame_stride, visualize_prompt
        self.frame_buffer_size, self.detect_interval = frame_buffer_size, detect_interval
        self.frame_buffer = []
        self.max_frame_num_to_track = max_frame_num_to_track
        self.max_inference_state_frames = max_inference_state_frames
        self.load_inference_state_path = load_inference_state_path
        self.save_inference_state_path = save_inference_state_path
        self.pre_frames = 0
        if save_inference_state_path is not None:
            assert max_inference_state_frames == -1, \
                "saving a preload memory bank requires max_inference_state_frames == -1 (det_sam2_RT.py:67-68)"
        if vis_frame_stride != -1 or visualize_prompt:
            raise NotImplementedError("rendering / prompt visualisation is outside the hot path")
        self.special_classes = 11
        self.special_classes_detection, self.special_classes_count = [], 0
        self.predictor = predictor or build_sam2_video_predictor(model_cfg, sam2_checkpoint, device=device)
        if detector is None:
            if detect_model_weights is None:
                raise ValueError("pass `detector` (callable frame_idx, frame_rgb -> detections) or YOLO weights")
            from ultralytics import YOLO  # third-party, out of scope; only used if installed

            yolo = YOLO(detect_model_weights)

            def detector(frame_idx, frame_rgb, _yolo=yolo):
                res = next(iter(_yolo([frame_rgb[..., ::-1]], stream=True, conf=self.detect_confidence, iou=0.1, verbose=False)))
                return [{"coordinates": b.xyxy[0].cpu().numpy(), "class": b.cls.cpu().numpy(), "confidence": b.conf.cpu().numpy()}
                        for b in (res.boxes or [])]
        self.detector = detector
        self.video_segments = {}
        self.inference_state = None
        self.pass_log = []
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)

    def clear(self):
        """det_sam2_RT.py:189-198."""
        self.frame_buffer, self.pre_frames = [], 0
        self.special_classes_detection, self.video_segments, self.inference_state = [], {}, None

    # ------------------------------------------------------------------ A2
    def detect_predict(self, images, past_num_frames):
        """det_sam2_RT.py:201-265."""
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
            cls = [int(np.asarray(d["class"]).reshape(-1)[0]) for d in dets]
            n_special = sum(1 for c in cls if c == self.special_classes)
            if n_special > self.special_classes_count:
                self.special_classes_detection = [d["coordinates"] for d, c in zip(dets, cls) if c == self.special_classes]
                self.special_classes_count = n_special
            res[f"frame_{t}"] = dets
        return res

    def Detect_2_SAM2_Prompt(self, detection_results_json):
        """det_sam2_RT.py:267-316."""
        for key, dets in (detection_results_json or {}).items():
            t = int(key.replace("frame_", ""))
            for d in dets:
                c = int(np.asarray(d["class"]).reshape(-1)[0])
                if c in self.skip_classes:
                    continue
                self.predictor.add_new_points_or_box(inference_state=self.inference_state, frame_idx=t, obj_id=c,
                                                     box=np.array(d["coordinates"], dtype=np.float32))
        return self.inference_state

    # ------------------------------------------------------------------ A1
    def Detect_and_SAM2_inference(self, frame_idx):
        """det_sam2_RT.py:342-411."""
        past = self.inference_state["num_frames"] if self.inference_state else 0
        dets = self.detect_predict(self.frame_buffer, past)
        self._ingest_buffer()
        self._prompt_and_propagate(frame_idx, dets)
        self._release(frame_idx)
        self._log_pass(frame_idx)

    def _ingest_buffer(self):
        """init_state / update_state with the buffered frames (det_sam2_RT.py:357-366)."""
        if self.inference_state is None:
            self.inference_state = self.predictor.init_state(video_path=self.frame_buffer)
        else:
            self.inference_state = self.predictor.update_state(video_path=self.frame_buffer, inference_state=self.inference_state)

    def _prompt_and_propagate(self, frame_idx, dets):
        """prompts + reverse propagation + threshold/host copy of one pass (det_sam2_RT.py:368-399)."""
        self.inference_state = self.Detect_2_SAM2_Prompt(dets)
        self._yielded, packed = [], []
        for t, obj_ids, bits in self.predictor.propagate_in_video(
                self.inference_state, start_frame_idx=frame_idx, max_frame_num_to_track=self.max_frame_num_to_track,
                reverse=True, output="packed"):
            self._yielded.append(t)
            if t >= self.pre_frames:
                packed.append((t, list(obj_ids), bits))
        # one device->host transfer per pass; each frame's masks stay bit-packed behind the reference's
        # {obj_id: bool[1,Hv,Wv]} mapping interface (PackedMasks)
        wv = self.inference_state["video_width"]
        if packed:
            host = torch.stack([b for _, _, b in packed]).cpu().numpy()
            for (t, ids, _), pb in zip(packed, host):
                self.video_segments[t] = PackedMasks(pb, ids, wv)

    def _release(self, frame_idx):
        """det_sam2_RT.py:404-411."""
        if self.max_inference_state_frames != -1:
            self.predictor.release_old_frames(self.inference_state, frame_idx, self.max_inference_state_frames,
                                              self.pre_frames, release_images=(self.vis_frame_stride == -1))

    def _log_pass(self, frame_idx):
        od = self.inference_state["output_dict"]
        self.pass_log.append((frame_idx, list(getattr(self, "_yielded", [])), sorted(od["cond_frame_outputs"]),
                              sorted(od["non_cond_frame_outputs"])))
        self._yielded = []

    def process_frame(self, frame_idx, frame):
        """det_sam2_RT.py:421-435."""
        self.frame_buffer.append(frame)
        if len(self.frame_buffer) >= self.frame_buffer_size:
            self.Detect_and_SAM2_inference(frame_idx)
            self.frame_buffer.clear()
        return self.inference_state

    # ------------------------------------------------------------------ A18 (preload bank)
    def save_inference_state(self, save_path):
        """det_sam2_RT.py:489-497.  The reference pickles the whole state (frames, devices, views); here the bank goes
        into a versioned tensor file (bank_io.DS2BANK: cond / non-cond entries + object table + the level-2 feature of
        every conditioning frame; no pickle, no frames)."""
        os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
        st = self.inference_state

        def fpn2_of(t):
            if t in st.get("preload_fpn2", {}):
                return st["preload_fpn2"][t]
            if t in st["images_idx"] or t in st["cached_features"]:
                return self.predictor._get_image_feature(st, t)[2]
            return None

        bank_io.save_bank(save_path, st, self.predictor.cfg.name, fpn2_of)

    def load_inference_state(self, load_path):
        """det_sam2_RT.py:499-503: DS2BANK files, and inference-state pickles written by the reference (read through a
        restricted unpickler and converted to the token-major layout).  Returns a host-resident state;
        ``init_preloading_state`` moves it to the GPU."""
        return bank_io.load_bank(load_path)

    def load_frames_from_folder(self, folder_path):
        """det_sam2_RT.py:507-524: the .png / .jpg / .jpeg files of a folder in sorted order, as a LIST-LIKE of RGB uint8
        arrays (``len()``, indexing, iteration - the reference returns a list).  The reference decodes with cv2.imread +
        BGR->RGB; here PIL decodes (PNG: identical pixels; JPEG: both sit on libjpeg, but the decoders' IDCT / chroma
        choices are not pinned against each other offline), EXIF orientation applied as cv2.imread does, and a frame is
        decoded when it is accessed, so a long folder never sits in host memory at once.  Files PIL cannot open are
        dropped up front (:516-518), so an all-unreadable folder is empty and ``run`` returns early as the reference does."""
        return _FolderFrames(folder_path)

    def _video_frames(self, video_path):
        """det_sam2_RT.py:558-579: cv2.VideoCapture + BGR->RGB.  OpenCV is an optional dependency of this path only."""
        try:
            import cv2
        except ImportError as e:
            raise NotImplementedError(
                "run(video_path=...) needs OpenCV (cv2.VideoCapture), which is not installed; extract the frames to a folder "
                "(ffmpeg -i <video> -q:v 2 -start_number 0 <dir>/%05d.jpg) and use run(frame_dir=...), or pass frames=") from e
        cap = cv2.VideoCapture(video_path)
        if not cap.isOpened():
            print(f"--- cannot open video file: {video_path}")
            return None

        def gen():
            try:
                while True:
                    ret, frame = cap.read()
                    if not ret:
                        return
                    yield cv2.cvtColor(frame, cv2.COLOR_BGR2RGB)
            finally:
                cap.release()
        return gen()

    def run(self, video_path=None, frame_dir=None, output_video_segments_pkl_path=None,
            output_special_classes_detection_pkl_path=None, frames=None):
        """det_sam2_RT.py:526-626.  Frame sources: ``video_path`` (cv2.VideoCapture, :558-579 - needs OpenCV),
        ``frame_dir`` (folder of .png / .jpg / .jpeg, :580-598) or ``frames`` (an in-memory iterable of HxWx3 uint8 RGB
        arrays: what both branches reduce to)."""
        if frames is None and video_path is not None:
            frames = self._video_frames(video_path)
            if frames is None:
                return None
        elif frames is None and frame_dir is not None:
            frames = self.load_frames_from_folder(frame_dir)
            if not len(frames):
                print(f"--- no frame files found in: {frame_dir}")
                return None
        elif frames is None:
            print("--- no video, frame folder or frames given")       # :600-601
            return None
        if self.load_inference_state_path is not None:
            self.inference_state = self.load_inference_state(self.load_inference_state_path)
            od = self.inference_state["output_dict"]
            self.inference_state["preloading_memory_cond_frame_idx"] = list(od["cond_frame_outputs"].keys())
            self.inference_state["preloading_memory_non_cond_frames_idx"] = list(od["non_cond_frame_outputs"].keys())
            self.pre_frames = self.inference_state["num_frames"]
            self.predictor.init_preloading_state(self.inference_state)
        idx = 0
        for fr in frames:
            self.inference_state = self.process_frame(self.pre_frames + idx, fr)
            idx += 1
        if self.frame_buffer:
            self.Detect_and_SAM2_inference(frame_idx=self.pre_frames + idx - 1)
            self.frame_buffer.clear()
        self.video_segments = {t - self.pre_frames: s for t, s in self.video_segments.items() if t >= self.pre_frames}
        if output_video_segments_pkl_path:    # plain {frame: {obj_id: bool[1,Hv,Wv]}} as the reference writes it (:612-615)
            with open(output_video_segments_pkl_path, "wb") as f:
                pickle.dump({t: (s.to_dict() if isinstance(s, PackedMasks) else s) for t, s in self.video_segments.items()}, f)
        if output_special_classes_detection_pkl_path and self.special_classes_detection is not None:
            with open(output_special_classes_detection_pkl_path, "wb") as f:
                pickle.dump(self.special_classes_detection, f)
        if self.save_inference_state_path is not None:
            self.save_inference_state(self.save_inference_state_path)
        return self.video_segments
