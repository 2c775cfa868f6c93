"""Helper of tests/test_hip_switches.py (run in a subprocess with one non-default switch in the environment): config 1 and the
16-object scenario on the tiny model against their reference goldens; prints the worst 1 - IoU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from det_sam2_amd.config import resolve_config  # noqa: E402
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame  # noqa: E402
from det_sam2_amd.weights import synthetic_state_dict  # noqa: E402

TINY = "sam2.1_hiera_t"


def iou(a, b):
    inter, union = np.logical_and(a, b).sum(), np.logical_or(a, b).sum()
    return 1.0 if union == 0 else inter / union


def main():
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    from oracle.make_goldens import B16_KW
    cfg = resolve_config(TINY)
    sd = synthetic_state_dict(cfg, 0)
    gold = os.path.join(ROOT, "tests", "golden")
    worst = 0.0
    # config 1: 8 frames, 1 box
    g = np.load(os.path.join(gold, "e2e_cfg1.npz"))
    pred = SAM2VideoPredictor(cfg, sd, "cuda:0", max_batch=16)
    vp = VideoProcessor(model_cfg=TINY, detector=SyntheticDetector(1), skip_classes=set(), predictor=pred, frame_buffer_size=8,
                        detect_interval=8, max_frame_num_to_track=8, max_inference_state_frames=-1)
    for t in range(8):
        vp.process_frame(t, synthetic_frame(t))
    for i, t in enumerate(g["frames"]):
        ref = np.unpackbits(g["bits"][i]).reshape(1, 1024, 1024).astype(bool)
        worst = max(worst, 1.0 - iou(vp.video_segments[int(t)][0], ref))
    # 16 objects, 3 frames
    g = np.load(os.path.join(gold, "e2e_b16.npz"))
    vp = VideoProcessor(model_cfg=TINY, detector=SyntheticDetector(16), predictor=SAM2VideoPredictor(cfg, sd, "cuda:0", max_batch=16),
                        **B16_KW)
    lows = []
    orig = vp.predictor.propagate_in_video

    def capture(st, **k):
        for t, ids, bits in orig(st, **k):
            od = st["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            lows.append((t, len(ids), od[key][t]["pred_masks"].clone()))
            yield t, ids, bits

    vp.predictor.propagate_in_video = capture
    for t in range(3):
        vp.process_frame(t, synthetic_frame(t))
    for i, (t, nobj, low) in enumerate(lows):
        low = low.cpu().numpy()
        ref_bits = np.unpackbits(g[f"lowbits{i}"])[: low.size].reshape(low.shape).astype(bool)
        for o in range(nobj):
            worst = max(worst, 1.0 - iou(low[o] > 0, ref_bits[o]))
    print("WORST", worst)


if __name__ == "__main__":
    main()
