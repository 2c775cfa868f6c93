"""Continuous companion of tools/prec_emulate.py: RMS / max difference of the low-res logits between the oracle run with an emulated
product scheme and the oracle run in exact fp32, on the 16-object tiny scenario (3 frames).  1 - IoU counts a handful of flipped
pixels; the RMS logit error ranks schemes without that quantisation.

    python tools/prec_rms.py bf16x3 mx_f16 sx_f16 ...      [DS2_EMU_ONLY=enc:s3 etc. as in prec_emulate.py]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import prec_emulate as PE  # noqa: E402
from det_sam2_amd.config import resolve_config  # noqa: E402
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame  # noqa: E402
from det_sam2_amd.weights import synthetic_state_dict  # noqa: E402
from oracle.video_processor import OracleVideoProcessor  # noqa: E402


def run(scheme, name, nobj, nframes):
    from oracle.make_goldens import B16_KW
    PE.SCHEME = scheme
    cfg = resolve_config(name)
    PE._dims(cfg)
    vp = OracleVideoProcessor(synthetic_state_dict(cfg, 1), cfg, SyntheticDetector(nobj), **B16_KW)
    with torch.inference_mode():
        for t in range(nframes):
            vp.process_frame(t, synthetic_frame(t, structured=True))
    od = vp.inference_state["output_dict"]
    out = []
    for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
        for t in sorted(od[key]):
            out.append(od[key][t]["pred_masks"].numpy().copy())
    return np.stack(out)


_orig_sdpa = F.scaled_dot_product_attention
HATT = [None]          # DS2_RMS_HATT=<scheme>: the attention cores of the IMAGE ENCODER (hieradet.py:40-82) in a reduced-term arithmetic,
                       # Linear layers as PE.SCHEME says: f16x1 (q, k, p, v one fp16 plane each: 2 MFMAs instead of 6), f16_pv2 (v two fp16
                       # planes: 3), f16_qk2 (k two planes: 3), bf16x1 (the rejected HQK1 family, calibration)


def emu_sdpa(q, k, v, *a, **kw):
    if HATT[0] is None or PE.REGION[0] != "enc" or a or kw:
        return _orig_sdpa(q, k, v, *a, **kw)
    sch = HATT[0]
    r = (lambda x: x.to(torch.bfloat16).float()) if sch == "bf16x1" else (lambda x: x.to(torch.float16).float())
    scale = q.shape[-1] ** -0.5
    qs = q * scale
    qh, kh = r(qs), r(k)
    s_ = qh @ kh.transpose(-1, -2)
    if sch == "f16_qk2":
        s_ = s_ + qh @ r(k - kh).transpose(-1, -2)
    p = torch.softmax(s_, dim=-1)
    ph, vh = r(p), r(v)
    o = ph @ vh
    if sch in ("f16_pv2", "f16_qk2"):
        o = o + ph @ r(v - vh)
    return o


if __name__ == "__main__":
    F.scaled_dot_product_attention = emu_sdpa
    HATT[0] = os.environ.get("DS2_RMS_HATT")
    F.linear = PE.emu_linear
    import oracle.modeling as _M
    for _n, _t in (("forward_image", "enc"), ("memory_attention", "ma"), ("mask_decoder", "dec"), ("memory_encoder", "menc")):
        PE._scoped(_M, _n, _t)
    torch.set_num_threads(int(os.environ.get("DS2_EMU_THREADS", "3")))
    name = os.environ.get("DS2_RMS_MODEL", "sam2.1_hiera_t")
    ref = run("exact", name, 16, 3)
    print(f"# {name}, 16 objects, 3 frames, weight seed 1, structured frames; logits rms {np.sqrt((ref ** 2).mean()):.3f}; DS2_EMU_ONLY={PE.ONLY}", flush=True)
    for sch in sys.argv[1:]:
        if ":" in sch:                 # <linear scheme>:<hiera attention scheme>
            sch, HATT[0] = sch.split(":")
        else:
            HATT[0] = os.environ.get("DS2_RMS_HATT")
        got = run(sch, name, 16, 3)
        d = got - ref
        flips = int(((got > 0) != (ref > 0)).sum())
        sch = sch + (":" + HATT[0] if HATT[0] else "")
        print(f"{sch:18s} rms dlogit {np.sqrt((d ** 2).mean()):.3e}   max {np.abs(d).max():.3e}   sign flips {flips} of {ref.size}", flush=True)
