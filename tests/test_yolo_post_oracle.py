"""Known-answer tests of oracle/yolo_post.py (the CPU restatement of ultralytics' non_max_suppression + scale_boxes that
csrc/detector_post.hip is held bit-exact to).  ultralytics / torchvision are absent offline (parity unpinned), so the
restatement is checked against hand-computed cases and an independent O(n^2) formulation of greedy NMS."""
import numpy as np

from oracle import yolo_post as Y


def _pred(boxes_xywh, scores):
    """boxes [N,4] xywh, scores [N,nc] -> [1, 4+nc, N]"""
    return np.concatenate([np.asarray(boxes_xywh, np.float32).T, np.asarray(scores, np.float32).T], 0)[None]


def test_hand_computed_case():
    # three boxes of class 0: A and B overlap (IoU 0.68), C is apart; D has class 1 and sits exactly on A
    boxes = [[100, 100, 50, 50], [105, 100, 50, 50], [300, 300, 40, 40], [100, 100, 50, 50], [10, 10, 4, 4]]
    scores = [[0.9, 0.0], [0.95, 0.0], [0.88, 0.1], [0.2, 0.92], [0.5, 0.3]]          # last one is below the threshold
    (d,) = Y.yolo_postprocess(_pred(boxes, scores), 0.85, 0.1)
    # order: B (0.95), D (0.92, class 1: the 7680 offset keeps it clear of A / B), C (0.88); A (0.90) is suppressed by B
    assert d.shape == (3, 6)
    assert np.allclose(d[:, 4], [0.95, 0.92, 0.88]) and list(d[:, 5]) == [0.0, 1.0, 0.0]
    assert np.array_equal(d[0, :4], np.array([80, 75, 130, 125], np.float32))
    assert np.array_equal(d[2, :4], np.array([280, 280, 320, 320], np.float32))


def test_iou_threshold_is_strict_and_ties_follow_anchor_order():
    # two identical-score boxes with IoU exactly 1/3: kept at iou_thres = 1/3 (suppression needs IoU > thr), the earlier
    # anchor first; dropped at a lower threshold
    boxes = [[10, 10, 20, 20], [20, 10, 20, 20]]
    scores = [[0.9], [0.9]]
    (d,) = Y.yolo_postprocess(_pred(boxes, scores), 0.5, float(np.float32(1) / np.float32(3)))
    assert d.shape[0] == 2 and d[0, 0] == 0 and d[1, 0] == 10
    (d,) = Y.yolo_postprocess(_pred(boxes, scores), 0.5, 0.3)
    assert d.shape[0] == 1 and d[0, 0] == 0


def test_against_an_independent_formulation_and_max_det():
    rng = np.random.default_rng(0)
    N, nc = 600, 5
    xy = rng.uniform(50, 590, (N, 2))
    wh = rng.uniform(20, 120, (N, 2))
    scores = rng.uniform(0, 1, (N, nc)) ** 3
    pred = _pred(np.concatenate([xy, wh], 1), scores)
    (d,) = Y.yolo_postprocess(pred, 0.3, 0.45, max_det=40)
    # independent: sort, then mark suppressed with a full pairwise IoU matrix in float64 (margins are far from fp32 noise here)
    conf, cls = scores.max(1), scores.argmax(1)
    idx = np.nonzero(conf.astype(np.float32) > np.float32(0.3))[0]
    order = idx[np.lexsort((idx, -conf[idx].astype(np.float32).astype(np.float64)))]
    b = np.concatenate([xy - wh / 2, xy + wh / 2], 1) + (cls * 7680.0)[:, None]
    keep = []
    for i in order:
        ok = True
        for k in keep:
            iw = max(min(b[i, 2], b[k, 2]) - max(b[i, 0], b[k, 0]), 0)
            ih = max(min(b[i, 3], b[k, 3]) - max(b[i, 1], b[k, 1]), 0)
            inter = iw * ih
            if inter / ((b[i, 2] - b[i, 0]) * (b[i, 3] - b[i, 1]) + (b[k, 2] - b[k, 0]) * (b[k, 3] - b[k, 1]) - inter) > 0.45:
                ok = False
                break
        if ok:
            keep.append(i)
    assert len(keep) > 40 and d.shape[0] == 40
    assert list(d[:, 5].astype(int)) == list(cls[keep[:40]])
    assert np.allclose(d[:, 4], conf[keep[:40]], rtol=1e-6)


def test_scale_back_to_the_original_image():
    gain, px, py = Y.letterbox_params((640, 640), (1080, 1920))       # 1920x1080 frame letterboxed into 640x640
    assert np.isclose(gain, 1 / 3) and (px, py) == (0, 140)
    boxes = [[320, 320, 100, 60], [5, 150, 30, 30]]
    scores = [[0.9], [0.8]]
    (d,) = Y.yolo_postprocess(_pred(boxes, scores), 0.5, 0.1, scale=(gain, px, py, 1920, 1080))
    assert np.allclose(d[0, :4], [810, 450, 1110, 630], atol=1e-3)
    assert d[1, 0] == 0.0 and d[1, 1] == 0.0                      # (-30, -15) clipped at the border
    assert (d[:, [0, 2]] >= 0).all() and (d[:, [0, 2]] <= 1920).all() and (d[:, [1, 3]] >= 0).all() and (d[:, [1, 3]] <= 1080).all()
