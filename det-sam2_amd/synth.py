"""Synthetic workload of SURVEY.md section 8d (no datasets, checkpoints or YOLO weights exist
offline): seeded 1024x1024 RGB frames and YOLO-contract detections on a 4x4 grid."""
from __future__ import annotations

import numpy as np


def synthetic_frame(frame_idx: int, seed: int = 0, size: int = 1024, structured: bool = False) -> np.ndarray:
    """uint8 [size,size,3] RGB. Default: uniform noise, ``default_rng(seed+frame_idx)``.
    ``structured``: moving discs on a gradient (visually meaningful masks)."""
    rng = np.random.default_rng(seed + frame_idx)
    if not structured:
        return rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    img = np.stack([xx / size * 160 + 40, yy / size * 160 + 40, (xx + yy) / (2 * size) * 120 + 60], -1)
    for o in range(16):
        cx, cy, s = _grid_box_center(o, size)
        cx += 3.0 * frame_idx * np.cos(o)
        cy += 3.0 * frame_idx * np.sin(o)
        m = (xx - cx) ** 2 + (yy - cy) ** 2 < (0.45 * s) ** 2
        img[m] = np.array([(37 * o) % 256, (91 * o + 50) % 256, (53 * o + 120) % 256], np.float32)
    img += rng.normal(0, 4.0, img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def _grid_box_center(o: int, size: int):
    cell = size / 4.0
    s = 180.0 * size / 1024.0
    gx, gy = o % 4, (o // 4) % 4
    return gx * cell + cell / 2, gy * cell + cell / 2, s


def synthetic_box(obj: int, frame_idx: int, seed: int = 0, size: int = 1024) -> np.ndarray:
    """float32 xyxy: ~180 px square in cell ``obj`` of a 4x4 grid, jittered +-8 px per detection frame."""
    cx, cy, s = _grid_box_center(obj, size)
    rng = np.random.default_rng([seed, 7919, obj, frame_idx])
    j = rng.uniform(-8, 8, 4).astype(np.float32)
    return np.array([cx - s / 2 + j[0], cy - s / 2 + j[1], cx + s / 2 + j[2], cy + s / 2 + j[3]], np.float32)


class SyntheticDetector:
    """Callable ``(frame_abs_idx, frame_rgb) -> list[detection dict]`` at the YOLO output contract
    (det_sam2_RT.py:228-244).  ``appear`` maps object id -> first frame on which it is detected; ``duplicates``
    maps object id -> number of EXTRA detections of that class per frame (YOLO emitting several boxes of one class:
    every further box is a second prompt for the same object on the same frame, sam2_video_predictor.py:470-483).
    ``class_ids[o]`` is the YOLO class id reported for object ``o`` (default: ``o``)."""

    def __init__(self, num_objects: int, seed: int = 0, size: int = 1024, appear=None, duplicates=None, class_ids=None):
        self.num_objects, self.seed, self.size = num_objects, seed, size
        self.appear = dict(appear or {})
        self.duplicates = dict(duplicates or {})
        self.class_ids = list(class_ids) if class_ids is not None else list(range(num_objects))

    def __call__(self, frame_idx, frame=None):
        out = []
        for o in range(self.num_objects):
            if frame_idx < self.appear.get(o, 0):
                continue
            out.append({"coordinates": synthetic_box(o, frame_idx, self.seed, self.size),
                        "class": np.array([float(self.class_ids[o])], np.float32),
                        "confidence": np.array([0.99], np.float32)})
            for k in range(self.duplicates.get(o, 0)):   # same class, a differently jittered box
                out.append({"coordinates": synthetic_box(o, frame_idx + 1000 * (k + 1), self.seed, self.size),
                            "class": np.array([float(self.class_ids[o])], np.float32),
                            "confidence": np.array([0.9], np.float32)})
        return out
