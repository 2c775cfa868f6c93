"""oracle/resize.py (restated OpenCV 8-bit INTER_LINEAR, parity-unpinned: cv2 is absent) - internal consistency:
identity, constants, agreement with real-arithmetic bilinear to within the fixed-point rounding, C-library tables."""
import numpy as np

from oracle.resize import cv2_resize_linear_u8, linear_coeffs


def _float_bilinear(img, dw, dh):
    H, W, _ = img.shape
    fx = (np.arange(dw) + 0.5) * W / dw - 0.5
    fy = (np.arange(dh) + 0.5) * H / dh - 0.5
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    wx, wy = fx - x0, fy - y0
    wx = np.where((x0 < 0) | (x0 >= W - 1), 0.0, wx)
    x0c, x1c = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    y0c, y1c = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    a = img.astype(np.float64)
    top = a[y0c][:, x0c] * (1 - wx)[None, :, None] + a[y0c][:, x1c] * wx[None, :, None]
    bot = a[y1c][:, x0c] * (1 - wx)[None, :, None] + a[y1c][:, x1c] * wx[None, :, None]
    return top * (1 - wy)[:, None, None] + bot * wy[:, None, None]


def test_resize_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    assert np.array_equal(cv2_resize_linear_u8(img, (480, 270)), img)                      # same size: copy
    for (dw, dh) in [(256, 256), (1024, 1024), (100, 333)]:                                 # down, up, mixed
        out = cv2_resize_linear_u8(img, (dw, dh))
        assert out.shape == (dh, dw, 3) and out.dtype == np.uint8
        assert np.abs(out.astype(np.float64) - _float_bilinear(img, dw, dh)).max() <= 1.0   # 11-bit weights + 2 shifts
    c = np.full((37, 91, 3), 201, np.uint8)
    assert np.unique(cv2_resize_linear_u8(c, (64, 48))).tolist() == [201]
    s, w0, w1 = linear_coeffs(1024, 1920, True)
    assert (w0 + w1 == 2048).sum() >= 1000 and s.min() == 0 and s.max() <= 1919 and np.all(np.diff(s) >= 0)
