"""F4 detector half on the GPU: csrc/detector_post.hip (through torch.ops.det_sam2.yolo_postprocess / the C-ABI) BIT-EXACT
against oracle/yolo_post.py - random head outputs incl. exact score ties, several images, many classes, letterbox undo,
max_det truncation, empty images, and the candidate-overflow flag."""
import numpy as np
import pytest
import torch

from oracle import yolo_post as Y

pytestmark = pytest.mark.gpu


def _random_pred(rng, nb, nc, N, quant=None, hot=1500):
    """Background scores < 0.05 everywhere; `hot` random anchors per image carry one confident class (so that the
    candidate set stays in the low thousands - the oracle's NMS is a Python loop)."""
    xy = rng.uniform(0, 640, (nb, 2, N))
    wh = rng.uniform(8, 200, (nb, 2, N))
    sc = rng.uniform(0, 0.05, (nb, nc, N))
    for b in range(nb):
        a = rng.choice(N, size=min(hot, N), replace=False)
        sc[b, rng.integers(0, nc, a.size), a] = rng.uniform(0.1, 1.0, a.size)
    if quant:                                   # exact ties between confidences
        sc = np.round(sc * quant) / quant
    return np.concatenate([xy, wh, sc], 1).astype(np.float32)


@pytest.mark.parametrize("nb,nc,N,conf,iou,max_det,quant", [
    (1, 1, 8400, 0.25, 0.45, 300, None),
    (3, 20, 8400, 0.30, 0.10, 300, None),       # the reference's iou = 0.1 (det_sam2_RT.py:228)
    (2, 80, 21504, 0.20, 0.45, 50, 64),         # 1024^2 input, COCO classes, ties, max_det truncation
    (2, 4, 2000, 0.999999, 0.5, 300, None),     # (almost) nothing over the threshold
])
def test_yolo_postprocess_bit_exact(nb, nc, N, conf, iou, max_det, quant):
    from det_sam2_amd import _capi
    ops = _capi.load_torch_ops()
    rng = np.random.default_rng(nb * 1000 + nc)
    pred = _random_pred(rng, nb, nc, N, quant)
    gain, px, py = Y.letterbox_params((640, 640), (1080, 1920))
    scale = (gain, px, py, 1920.0, 1080.0)
    want = Y.yolo_postprocess(pred, conf, iou, max_det, scale)
    st = torch.tensor(scale, dtype=torch.float32, device="cuda:0")
    dets, counts = ops.yolo_postprocess(torch.from_numpy(pred).to("cuda:0"), conf, iou, max_det, st)
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    for b in range(nb):
        assert counts[b] == want[b].shape[0], (b, counts[b], want[b].shape)
        assert np.array_equal(dets[b, : counts[b]].view(np.uint32), want[b].view(np.uint32)), b     # bit-exact
    assert sum(w.shape[0] for w in want) > 0 or conf > 0.99


def test_python_wrapper_emits_the_detection_contract_and_flags_overflow():
    from det_sam2_amd.detector import HeadDetector, yolo_postprocess
    rng = np.random.default_rng(5)
    pred = torch.from_numpy(_random_pred(rng, 1, 3, 8400)).to("cuda:0")
    frame = np.zeros((1080, 1920, 3), np.uint8)
    det = HeadDetector(lambda f: (pred, (640, 640)), conf=0.5, iou=0.1)
    out = det(0, frame)
    ref = Y.yolo_postprocess(pred.cpu().numpy(), 0.5, 0.1, 300, (*Y.letterbox_params((640, 640), (1080, 1920)), 1920.0, 1080.0))[0]
    assert len(out) == ref.shape[0] > 0
    for d, r in zip(out, ref):
        assert d["coordinates"].dtype == np.float32 and d["coordinates"].shape == (4,) and d["class"].shape == (1,)
        assert np.array_equal(d["coordinates"], r[:4]) and d["class"][0] == r[5] and d["confidence"][0] == r[4]
    with pytest.raises(RuntimeError, match="8192"):                  # every anchor over the threshold: more than the NMS stage holds
        yolo_postprocess(torch.from_numpy(_random_pred(rng, 1, 1, 20000, hot=10)).to("cuda:0") + 10.0, 0.0)
    with pytest.raises(RuntimeError, match="GPU"):
        yolo_postprocess(pred.cpu(), 0.5)
