"""Diagnostic for tests/test_hip_measured_shape.py: per frame / object 1 - IoU, mask area and logit error of the hiera_l x 16
run against the reference golden, per arithmetic mode.   python tools/diag_large_b16.py [s1|seed0] [modes...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from det_sam2_amd.config import resolve_config  # noqa: E402
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame  # noqa: E402
from det_sam2_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    from oracle.make_goldens import HELDOUT, L16_FRAMES, L16_KW
    variant = sys.argv[1] if len(sys.argv) > 1 else "s1"
    modes = sys.argv[2:] or ["fp32", "bf16x3", "bf16x3k"]
    ws, ls, st = (0, 1.0, False) if variant == "seed0" else HELDOUT[variant]
    name = os.environ.get("DIAG_MODEL", "sam2.1_hiera_l")
    nobj = int(os.environ.get("DIAG_OBJ", "16"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_large_b16.npz" if variant == "seed0" else f"ho_large_b16_{variant}.npz"))
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, ws, ls)
    for mode in modes:
        pred = SAM2VideoPredictor(cfg, sd, "cuda:0", max_batch=nobj)
        pred.hip.set_precision(mode)
        vp = VideoProcessor(model_cfg=name, detector=SyntheticDetector(nobj), predictor=pred, **L16_KW)
        lows = []
        orig = vp.predictor.propagate_in_video

        def capture(state, **k):
            for t, ids, bits in orig(state, **k):
                od = state["output_dict"]
                key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
                lows.append((t, len(ids), od[key][t]["pred_masks"].clone()))
                yield t, ids, bits

        vp.predictor.propagate_in_video = capture
        for t in range(L16_FRAMES):
            vp.process_frame(t, synthetic_frame(t, structured=st))
        print(f"== {variant} {mode}")
        for i, (t, n, low) in enumerate(lows):
            low = low.cpu().numpy()
            ref_bits = np.unpackbits(g[f"lowbits{i}"])[: low.size].reshape(low.shape).astype(bool)
            sub, ref = low[:, :, ::4, ::4], g[f"low{i}"].astype(np.float32)
            rows = []
            for o in range(n):
                a, b = low[o] > 0, ref_bits[o]
                u = np.logical_or(a, b).sum()
                x = np.logical_xor(a, b).sum()
                rows.append((x / max(u, 1), int(x), int(u), float(np.abs(sub[o] - ref[o]).max()), float(np.abs(sub[o] - ref[o]).mean())))
            worst = max(r[0] for r in rows)
            print(f"frame {t}: worst 1-IoU {worst:.2e}  " + " ".join(f"[{r[1]}/{r[2]} d{r[3]:.1e}]" for r in rows), flush=True)
        del vp, pred


if __name__ == "__main__":
    main()
