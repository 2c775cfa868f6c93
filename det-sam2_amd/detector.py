"""Detector side of the stream loop (detector half): YOLOv8 head output -> the detection dicts
``VideoProcessor.detect_predict`` consumes (det_sam2_RT.py:228-244), on the GPU.

The reference calls ``ultralytics.YOLO(weights)(frames, conf=detect_confidence, iou=0.1)`` and reads ``result.boxes``; the
network itself and its weights are third-party and absent offline (OUT OF SCOPE).  What IS rebuilt is
the arithmetic between the head and the boxes - best class per anchor, confidence threshold, xywh -> xyxy, per-class NMS,
max_det, letterbox undo + clip (ultralytics ``ops.non_max_suppression / scale_boxes``; HIP: csrc/detector_post.hip through
``torch.ops.det_sam2.yolo_postprocess``), so a head exported from any framework can feed the tracker without a CUDA-only
dependency.  Parity with ultralytics is unpinned offline; the kernels are bit-exact against oracle/yolo_post.py.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _capi


def letterbox_params(net_hw, orig_hw):
    """gain and (pad_x, pad_y) of the letterbox that mapped the original frame onto the network input
    (ultralytics scale_boxes with ratio_pad=None)."""
    gain = min(net_hw[0] / orig_hw[0], net_hw[1] / orig_hw[1])
    return float(gain), float(round((net_hw[1] - orig_hw[1] * gain) / 2 - 0.1)), float(round((net_hw[0] - orig_hw[0] * gain) / 2 - 0.1))


def yolo_postprocess(pred: torch.Tensor, conf_thres: float, iou_thres: float = 0.1, max_det: int = 300, net_hw=None, orig_hw=None):
    """pred fp32 [nb, 4+nc, N] on the GPU -> list (one per image) of detection lists at the YOLO output contract:
    ``{"coordinates": float32[4] xyxy, "class": float32[1], "confidence": float32[1]}`` in confidence order."""
    if not pred.is_cuda:
        raise RuntimeError("yolo_postprocess: GPU tensor required (there is no CPU path)")
    ops = _capi.load_torch_ops()
    scale = None
    if net_hw is not None and orig_hw is not None:
        gain, px, py = letterbox_params(net_hw, orig_hw)
        scale = torch.tensor([gain, px, py, float(orig_hw[1]), float(orig_hw[0])], dtype=torch.float32, device=pred.device)
    dets, counts = ops.yolo_postprocess(pred.to(torch.float32).contiguous(), float(conf_thres), float(iou_thres), int(max_det), scale)
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    out = []
    for b in range(dets.shape[0]):
        if counts[b] < 0:
            raise RuntimeError("yolo_postprocess: more than 8192 anchors over the confidence threshold in one image")
        out.append([{"coordinates": d[:4].astype(np.float32), "class": d[5:6].astype(np.float32), "confidence": d[4:5].astype(np.float32)}
                    for d in dets[b, : counts[b]]])
    return out


class HeadDetector:
    """``detector=`` for VideoProcessor: ``(frame_idx, frame_rgb) -> [detection dict]``.  ``head(frame_rgb) -> (pred fp32
    [1, 4+nc, N] on the GPU, (net_h, net_w))`` is the caller's YOLOv8 network (pre-processing + backbone + head)."""

    def __init__(self, head, conf: float = 0.85, iou: float = 0.1, max_det: int = 300):
        self.head, self.conf, self.iou, self.max_det = head, conf, iou, max_det

    def __call__(self, frame_idx, frame_rgb):
        pred, net_hw = self.head(frame_rgb)
        return yolo_postprocess(pred, self.conf, self.iou, self.max_det, net_hw, frame_rgb.shape[:2])[0]
