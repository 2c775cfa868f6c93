"""TEST INFRASTRUCTURE: CPU restatement (numpy, fp32) of the detector post-processing that sits between the YOLOv8 head and
``VideoProcessor.detect_predict`` (det_sam2_RT.py:228-244 reads ``result.boxes`` -> xyxy / cls / conf).

That arithmetic lives in a third-party dependency that is ABSENT from /root/reference and from this image:
``ultralytics==8.2.82`` (requirements.txt) - ``ultralytics/utils/ops.py: non_max_suppression, xywh2xyxy, scale_boxes,
clip_boxes`` - on top of ``torchvision.ops.nms``.  **Parity unpinned**: neither package is installable offline and the
reference holds no fixture for this step; what follows restates the published algorithm with the defaults the reference's
call uses (``self.detect_model(frames, conf=detect_confidence, iou=0.1)``, det_sam2_RT.py:228: multi_label=False,
agnostic=False, max_det=300, max_nms=30000, max_wh=7680), and the HIP kernels (csrc/detector_post.hip) are held bit-exact
to it.  One choice the published code leaves to the sort implementation is fixed here: boxes of EQUAL confidence are
visited in ascending anchor order.

    pred  fp32 [nb, 4 + nc, N]   raw head output: box centre x, y, width, height (network pixels) + nc class scores
    ->    list over images of fp32 [n, 6]: x1, y1, x2, y2 (original-image pixels when `scale` is given), conf, cls
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
MAX_WH = F32(7680.0)


def letterbox_params(net_hw, orig_hw):
    """ultralytics scale_boxes (ratio_pad=None): gain and (pad_x, pad_y) of the letterbox that mapped orig -> net."""
    gain = min(net_hw[0] / orig_hw[0], net_hw[1] / orig_hw[1])
    pad_x = round((net_hw[1] - orig_hw[1] * gain) / 2 - 0.1)
    pad_y = round((net_hw[0] - orig_hw[0] * gain) / 2 - 0.1)
    return F32(gain), F32(pad_x), F32(pad_y)


def iou_f32(a, b):
    """torchvision nms kernel: inter / (area_a + area_b - inter), fp32, no fused multiply-add."""
    w = max(F32(min(a[2], b[2]) - max(a[0], b[0])), F32(0))
    h = max(F32(min(a[3], b[3]) - max(a[1], b[1])), F32(0))
    inter = F32(w * h)
    sa = F32(F32(a[2] - a[0]) * F32(a[3] - a[1]))
    sb = F32(F32(b[2] - b[0]) * F32(b[3] - b[1]))
    return F32(inter / F32(F32(sa + sb) - inter))


def nms_greedy(boxes, order, iou_thres):
    """Greedy NMS over `boxes` visited in `order`: a box is dropped if its IoU with an earlier KEPT box is > iou_thres.
    (iou_f32 applied to all kept boxes at once: the same fp32 operations element by element.)"""
    boxes = np.asarray(boxes, F32)
    area = ((boxes[:, 2] - boxes[:, 0]).astype(F32) * (boxes[:, 3] - boxes[:, 1]).astype(F32)).astype(F32)
    keep = []
    for i in order:
        if keep:
            k = boxes[keep]
            w = np.maximum((np.minimum(boxes[i, 2], k[:, 2]) - np.maximum(boxes[i, 0], k[:, 0])).astype(F32), F32(0))
            h = np.maximum((np.minimum(boxes[i, 3], k[:, 3]) - np.maximum(boxes[i, 1], k[:, 1])).astype(F32), F32(0))
            inter = (w * h).astype(F32)
            with np.errstate(divide="ignore", invalid="ignore"):
                iou = (inter / ((area[i] + area[keep]).astype(F32) - inter).astype(F32)).astype(F32)
            if (iou > F32(iou_thres)).any():
                continue
        keep.append(int(i))
    return keep


def yolo_postprocess(pred, conf_thres, iou_thres, max_det=300, scale=None):
    """``scale = (gain, pad_x, pad_y, orig_w, orig_h)`` maps the boxes back to the original image (scale_boxes + clip_boxes)."""
    pred = np.asarray(pred, F32)
    out = []
    for p in pred:
        cls_scores = p[4:]                                     # [nc, N]
        conf = cls_scores.max(0)
        cls = cls_scores.argmax(0)                             # first maximum
        cand = np.nonzero(conf > F32(conf_thres))[0]           # ascending anchor index
        if cand.size == 0:
            out.append(np.zeros((0, 6), F32))
            continue
        x, y, w, h = (p[i, cand] for i in range(4))
        hw, hh = (w / F32(2)).astype(F32), (h / F32(2)).astype(F32)      # xywh2xyxy
        box = np.stack([x - hw, y - hh, x + hw, y + hh], 1).astype(F32)
        off = (cls[cand].astype(F32) * MAX_WH)[:, None]                  # per-class offset: classes never suppress each other
        order = np.lexsort((cand, -conf[cand].astype(np.float64)))       # conf descending, ties by ascending anchor
        keep = nms_greedy((box + off).astype(F32), order, iou_thres)[:max_det]
        b = box[keep]
        if scale is not None:
            gain, px, py, ow, oh = scale
            b = b.copy()
            b[:, [0, 2]] = ((b[:, [0, 2]] - F32(px)) / F32(gain)).astype(F32)
            b[:, [1, 3]] = ((b[:, [1, 3]] - F32(py)) / F32(gain)).astype(F32)
            b[:, [0, 2]] = np.clip(b[:, [0, 2]], F32(0), F32(ow))
            b[:, [1, 3]] = np.clip(b[:, [1, 3]], F32(0), F32(oh))
        out.append(np.concatenate([b, conf[cand][keep, None], cls[cand][keep, None].astype(F32)], 1).astype(F32))
    return out
