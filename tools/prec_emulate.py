"""CPU experiment: would a product of TWO MFMA-equivalents instead of three pass the parity gate?

The bf16x3 GEMMs compute a.w = a_hi.w_hi + a_lo.w_hi + a_hi.w_lo on bf16 planes (3 MFMAs, error ~2^-17).  This script runs the
ORACLE (fp32 PyTorch-CPU restatement, test infrastructure) with every large Linear layer replaced by an emulation of a candidate
split-precision product and reports worst 1 - IoU / max |dlogit| against the reference goldens - the same gate the arithmetic
modes of section 4 of DESIGN.md were selected with (every golden <= 5e-4).

    python tools/prec_emulate.py <scheme> [cfg1|b16|stream2|large] ...

Schemes (cost in bf16-MFMA equivalents per product; MX-fp8 and int8 MFMAs run at twice the bf16 rate on gfx950):
    exact     fp32 (sanity: reproduces the goldens)
    bf16x3    today's product                                              3.0
    gemm2a    bf16x3 without the activation's lo plane (calibration: measured 6.1e-3 on cfg1 on the GPU)   2.0
    mx_bf16   bf16 hi.hi + MX-fp8 (e4m3, 32-element block scales) cross terms a_lo8.w_hi8 + a_hi8.w_lo8    2.0
    mx_f16    fp16 hi.hi + the same MX-fp8 cross terms                     2.0
    f16x3     fp16 planes, three terms                                     3.0
    f16x2a    fp16 hi.hi + a_lo.w_hi (weights rounded to fp16)             2.0
    f16x2w    fp16 hi.hi + a_hi.w_lo (activations rounded to fp16)         2.0
    f16x1     one fp16 plane each                                          1.0
    i8x3      per-row 16-bit fixed point as two int8 planes, hi.hi + both cross terms      1.5
    i8x4      the same with all four terms (exact 16-bit fixed-point product)             2.0
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from det_sam2_amd.config import resolve_config  # noqa: E402
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame  # noqa: E402
from det_sam2_amd.weights import synthetic_state_dict  # noqa: E402
from oracle.video_processor import OracleVideoProcessor  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
_orig_linear = F.linear
SCHEME = "exact"


def _bf16(x):
    return x.to(torch.bfloat16).float()


def _f16(x):
    return x.to(torch.float16).float()


def _mx8(x):
    """MX-fp8 (e4m3) with one power-of-two scale per 32 consecutive K elements (last dim)."""
    K = x.shape[-1]
    pad = (-K) % 32
    xp = F.pad(x, (0, pad)) if pad else x
    b = xp.reshape(*xp.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    q = (b / scale).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(xp.shape)[..., :K]


def _fix16(x):
    """per-row symmetric 16-bit fixed point: returns (q int-valued float, hi, lo, scale) with q = 256 hi + lo."""
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    s = amax / 32767.0
    q = torch.round(x / s)
    hi = torch.round(q / 256.0).clamp(-128, 127)
    lo = q - 256.0 * hi          # in [-128, 128]; (128 only when hi was clamped: negligible)
    return q, hi, lo, s


REGION = [None]         # which stage of the oracle is running (DS2_EMU_ONLY=enc|ma|dec|menc: the scheme there, bf16x3 elsewhere)
ONLY = os.environ.get("DS2_EMU_ONLY")


def _scoped(mod, name, tag):
    f = getattr(mod, name)

    def g(*a, **k):
        prev, REGION[0] = REGION[0], tag
        try:
            return f(*a, **k)
        finally:
            REGION[0] = prev
    setattr(mod, name, g)


ENC_DIMS = [None]       # [e, 2e, 4e, 8e] of the model being run (set by the runners)


def _enc_group(w):
    """layer group of a Linear weight [N, K] inside forward_image: s12 (Hiera stages 1 - 2, every role) | s3qkv | s3proj | s3fc1 | s3fc2 |
    s4 (stage 4, every role) | neck (everything else: FPN 1x1 convolutions run as Linear layers, patch embedding)"""
    N, K = w.shape
    d = ENC_DIMS[0]
    role, dim = None, None
    if N == 4 * K:
        role, dim = "fc1", K
    elif K == 4 * N:
        role, dim = "fc2", N
    elif N % 3 == 0 and N // 3 in (K, 2 * K):
        role, dim = "qkv", K
    elif N in (K, 2 * K):
        role, dim = "proj", K
    if role is None or dim not in d:
        return "neck"
    st = d.index(dim)
    if st <= 1:
        return "s12"
    if st == 3 or (st == 2 and N == 2 * K and role == "proj") or (st == 2 and role == "qkv" and N // 3 == 2 * K):
        return "s4"          # (the dimension change into stage 4 belongs to its first block)
    return "s3" + role


def emu_linear(x, w, b=None):
    global SCHEME
    rows = x.numel() // x.shape[-1]
    if SCHEME == "exact" or rows <= 128:          # k_skinny_linear: exact fp32 in every mode
        return _orig_linear(x, w, b)
    in_region = REGION[0] == ONLY
    if ONLY == "encmlp":      # the MLP GEMMs of the Hiera blocks only (fc1: N = 4 K, fc2: K = 4 N)
        in_region = REGION[0] == "enc" and (w.shape[0] == 4 * w.shape[1] or w.shape[1] == 4 * w.shape[0])
    if ONLY == "encfc2":
        in_region = REGION[0] == "enc" and w.shape[1] == 4 * w.shape[0]
    if ONLY == "encfc1":
        in_region = REGION[0] == "enc" and w.shape[0] == 4 * w.shape[1]
    if ONLY and ONLY.startswith("enc:"):          # one layer group of the image encoder: enc:<group>
        in_region = REGION[0] == "enc" and (_enc_group(w) == ONLY[4:] or (ONLY[4:] == "s3" and _enc_group(w).startswith("s3"))
                                            or (ONLY[4:] == "s34" and _enc_group(w)[:2] in ("s3", "s4")))
    if ONLY and not in_region and SCHEME != "bf16x3":
        keep, SCHEME = SCHEME, "bf16x3"
        try:
            return emu_linear(x, w, b)
        finally:
            SCHEME = keep
    mm = lambda a, ww: _orig_linear(a, ww)        # noqa: E731  fp32 accumulation
    if SCHEME == "bf16x3":
        xh, wh = _bf16(x), _bf16(w)
        xl, wl = _bf16(x - xh), _bf16(w - wh)
        y = mm(xl, wh) + mm(xh, wl) + mm(xh, wh)
    elif SCHEME == "gemm2a":
        xh, wh = _bf16(x), _bf16(w)
        wl = _bf16(w - wh)
        y = mm(xh, wl) + mm(xh, wh)
    elif SCHEME in ("mx_bf16", "mx_f16"):
        r = _bf16 if SCHEME == "mx_bf16" else _f16
        xh, wh = r(x), r(w)
        y = mm(_mx8(x - xh), _mx8(w)) + mm(_mx8(x), _mx8(w - wh)) + mm(xh, wh)
    elif SCHEME in ("sx_f16", "sx_f16_t", "sx_f16_43"):
        # round 6: fp16 hi.hi + two fp8 cross terms with STATIC power-of-two scales (no block maxima in the producers): the hi operand of a
        # cross term is e5m2 = the fp16 plane's own top byte (rounded; `_t`: truncated, i.e. literally the byte; `_43`: e4m3 with the
        # per-tensor scale of the lo plane's tensor), the lo operand e4m3 of (x - x_hi) * 2^S, S from the tensor's absmax
        xh, wh = _f16(x), _f16(w)
        def lo8(d, ref):
            S = torch.floor(torch.log2(448.0 / (ref.abs().max().clamp_min(1e-30) * 2.0 ** -11)))       # lo <= 2^-11 |ref|
            return (d * torch.exp2(S)).clamp(-448, 448).to(torch.float8_e4m3fn).float() * torch.exp2(-S)
        def hi8(h):
            if SCHEME == "sx_f16_t":
                return (h.to(torch.float16).view(torch.int16) & -256).view(torch.float16).float()
            if SCHEME == "sx_f16_43":
                S = torch.floor(torch.log2(448.0 / h.abs().max().clamp_min(1e-30)))
                return (h * torch.exp2(S)).to(torch.float8_e4m3fn).float() * torch.exp2(-S)
            return h.clamp(-57344, 57344).to(torch.float8_e5m2).float()
        y = mm(lo8(x - xh, x), hi8(wh)) + mm(hi8(xh), lo8(w - wh, w)) + mm(xh, wh)
    elif SCHEME in ("mx_f16_a", "mx_f16_w"):      # fp16 hi.hi + ONE MX-fp8 cross term (1.5 equivalents)
        xh, wh = _f16(x), _f16(w)
        y = (mm(_mx8(x - xh), _mx8(w)) if SCHEME == "mx_f16_a" else mm(_mx8(x), _mx8(w - wh))) + mm(xh, wh)
    elif SCHEME in ("f16x3", "f16x2a", "f16x2w", "f16x1"):   # fp16 planes: 11-bit mantissas (range: |x| < 65504, subnormal below 6e-5)
        xh, wh = _f16(x), _f16(w)
        xl, wl = _f16(x - xh), _f16(w - wh)
        y = mm(xh, wh)
        if SCHEME in ("f16x3", "f16x2a"):
            y = y + mm(xl, wh)
        if SCHEME in ("f16x3", "f16x2w"):
            y = y + mm(xh, wl)
    elif SCHEME in ("i8x3", "i8x4"):
        qx, xh, xl, sx = _fix16(x)
        qw, wh, wl, sw = _fix16(w)
        acc = mm(qx.double(), qw.double())                      # exact integer arithmetic (int32 accumulators on the GPU)
        if SCHEME == "i8x3":
            acc = acc - mm(xl.double(), wl.double())
        y = (acc * sx.double() * sw.double().reshape(-1)).float()
    else:
        raise SystemExit(f"unknown scheme {SCHEME}")
    return y if b is None else y + b


def _iou(a, b):
    inter, union = np.logical_and(a, b).sum(), np.logical_or(a, b).sum()
    return 1.0 if union == 0 else inter / union


def _dims(cfg):
    e = cfg.trunk.embed_dim
    ENC_DIMS[0] = [e, 2 * e, 4 * e, 8 * e]


def run_cfg1():
    cfg = resolve_config("sam2.1_hiera_t")
    _dims(cfg)
    g = np.load(os.path.join(GOLD, "e2e_cfg1.npz"))
    vp = OracleVideoProcessor(synthetic_state_dict(cfg, 0), cfg, SyntheticDetector(1), skip_classes=set(), frame_buffer_size=8,
                              detect_interval=8, max_frame_num_to_track=8, max_inference_state_frames=-1)
    with torch.inference_mode():
        for t in range(8):
            vp.process_frame(t, synthetic_frame(t))
    od = vp.inference_state["output_dict"]
    worst, dl = 0.0, 0.0
    for i, t in enumerate(g["frames"]):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][int(t)]["pred_masks"].numpy()
        dl = max(dl, float(np.abs(low - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(1, 1024, 1024).astype(bool)
        worst = max(worst, 1.0 - _iou(vp.video_segments[int(t)][0], ref))
    return worst, dl


def _compact(gname, nobj, kw, nframes, name="sam2.1_hiera_t", variant=None):
    cfg = resolve_config(name)
    _dims(cfg)
    g = np.load(os.path.join(GOLD, gname))
    ws, ls, st = (0, 1.0, False)
    if variant:
        from oracle.make_goldens import HELDOUT
        ws, ls, st = HELDOUT[variant]
    vp = OracleVideoProcessor(synthetic_state_dict(cfg, ws, ls), cfg, SyntheticDetector(nobj), **kw)
    lows = []
    orig = vp.predictor.propagate_in_video

    def capture(st, **k):
        for t, ids, logits in orig(st, **k):
            od = st["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            lows.append((t, len(ids), od[key][t]["pred_masks"].clone().numpy()))
            yield t, ids, logits

    vp.predictor.propagate_in_video = capture
    with torch.inference_mode():
        for t in range(nframes):
            vp.process_frame(t, synthetic_frame(t, structured=st))
    worst, dl = 0.0, 0.0
    for i, (t, n, low) in enumerate(lows):
        ref = np.unpackbits(g[f"lowbits{i}"])[: low.size].reshape(low.shape).astype(bool)
        for o in range(n):
            worst = max(worst, 1.0 - _iou(low[o] > 0, ref[o]))
        dl = max(dl, float(np.abs(low[:, :, ::4, ::4] - g[f"low{i}"].astype(np.float32)).max()))
    return worst, dl


def run_b16():
    from oracle.make_goldens import B16_KW
    return _compact("e2e_b16.npz", 16, B16_KW, 3)


def run_ho_b16_lm():            # held-out: weight seed 1, structured frames, low-margin logits (the two most sensitive fixtures)
    from oracle.make_goldens import B16_KW
    return _compact("ho_b16_lm.npz", 16, B16_KW, 3, variant="lm")


def run_ho_large_b16_s1():      # hiera_l x 16 objects x 9 frames at the bench's bank (10+ minutes of CPU)
    from oracle.make_goldens import L16_FRAMES, L16_KW
    return _compact("ho_large_b16_s1.npz", 16, L16_KW, L16_FRAMES, name="sam2.1_hiera_l", variant="s1")


def run_large():
    from oracle.make_goldens import LARGE_KW
    cfg = resolve_config("sam2.1_hiera_l")
    _dims(cfg)
    g = np.load(os.path.join(GOLD, "e2e_large.npz"))
    vp = OracleVideoProcessor(synthetic_state_dict(cfg, 0), cfg, SyntheticDetector(2), **LARGE_KW)
    with torch.inference_mode():
        for t in range(3):
            vp.process_frame(t, synthetic_frame(t))
    od = vp.inference_state["output_dict"]
    worst, dl = 0.0, 0.0
    for i, t in enumerate(g["frames"]):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][int(t)]["pred_masks"].numpy()
        dl = max(dl, float(np.abs(low - g["low"][i]).max()))
        ref = np.unpackbits(g["bits"][i]).reshape(2, 1, 1024, 1024).astype(bool)
        for o in range(2):
            worst = max(worst, 1.0 - _iou(vp.video_segments[int(t)][o], ref[o]))
    return worst, dl


if __name__ == "__main__":
    SCHEME = sys.argv[1]
    F.linear = emu_linear
    import oracle.modeling as _M
    for _n, _t in (("forward_image", "enc"), ("memory_attention", "ma"), ("mask_decoder", "dec"), ("memory_encoder", "menc")):
        _scoped(_M, _n, _t)
    torch.set_num_threads(int(os.environ.get("DS2_EMU_THREADS", "2")))
    for case in (sys.argv[2:] or ["cfg1"]):
        worst, dl = globals()["run_" + case]()
        print(f"{SCHEME:8s} {case:8s} worst 1-IoU {worst:.3e}   max|dlogit| {dl:.3e}", flush=True)
