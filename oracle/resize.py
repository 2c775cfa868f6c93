"""ORACLE (test infrastructure only): restatement of ``cv2.resize(frame_rgb, (S, S))`` as the reference calls it in
``load_video_frames`` (sam2/utils/misc.py:328-334): uint8 HxWx3 input, default interpolation INTER_LINEAR.

**Parity unpinned**: OpenCV (third-party, version unpinned in the reference's requirements; not installed in this
image) cannot be run here and the reference holds no fixtures for it.  This restates the published algorithm of
OpenCV's imgproc/resize.cpp for 8-bit INTER_LINEAR (``resizeGeneric_`` with ``HResizeLinear<uchar,int,short,2048>``
and ``VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>``): pixel-centre mapping
``f = (d + 0.5) * (src/dst) - 0.5`` evaluated in double and stored as float, taps ``floor(f)`` / ``+1`` clamped to
the image, weights rounded to 11-bit fixed point (``cvRound(w * 2048)``, saturated to int16), horizontal pass in
int32, vertical pass ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``.  Builds of OpenCV that route
this call to IPP may differ by one grey level.  Same size => copy (cv::resize early-out).
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def linear_coeffs(dst: int, src: int, clamp_weights: bool):
    """Tap index and the two int16 weights per destination coordinate.  ``clamp_weights``: the x pass zeroes the
    fraction when the tap falls outside (resize.cpp: ``if (sx < 0) fx = 0, sx = 0`` / ``if (sx >= w-1) fx = 0,
    sx = w-1``); the y pass only clips the ROW INDICES and keeps the fraction."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_weights:
        lo, hi = s < 0, s >= src - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, src - 1, s)).astype(np.int32)
    w1 = np.clip(np.rint(f * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
    w0 = np.clip(np.rint((np.float32(1) - f) * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
    return s, w0, w1


def cv2_resize_linear_u8(img, dsize):
    """img uint8 [H,W,C]; dsize = (dst_width, dst_height) as in cv2.resize.  Returns uint8 [dst_h, dst_w, C]."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    dw, dh = int(dsize[0]), int(dsize[1])
    H, W, _ = img.shape
    if (H, W) == (dh, dw):
        return img.copy()
    sx, a0, a1 = linear_coeffs(dw, W, True)
    sy, b0, b1 = linear_coeffs(dh, H, False)
    x0, x1 = sx, np.minimum(sx + 1, W - 1)
    y0, y1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    src = img.astype(np.int32)
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]          # [H, dw, C] int32
    s0, s1 = rows[y0], rows[y1]                                                          # [dh, dw, C]
    v = ((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16)
    return np.clip((v + 2) >> 2, 0, 255).astype(np.uint8)
