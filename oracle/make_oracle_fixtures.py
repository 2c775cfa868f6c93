"""Fixtures whose expectation comes from the ORACLE (oracle/, pinned to the reference by tests/test_oracle_golden.py), for
behaviour the reference cannot produce on a CPU: hole filling.  ``fill_holes_in_mask_scores`` (sam2/utils/misc.py:365-393)
needs the reference's CUDA extension and silently returns its input without it (:389-391), so a fixture of the SHIPPING
default (build_sam.py:134 appends ``fill_hole_area=8``; sam2_video_predictor.py:1343-1346 applies it to every inferred
frame) cannot be recorded from /root/reference in this container.  The oracle restates that path with
``oracle/cc.py`` (8-connectivity labelling, scipy.ndimage.label contract) and is timed in minutes at the measured shape, so
its outputs are committed as a fixture instead of being recomputed on the GPU box.

    python -m oracle.make_oracle_fixtures fill8_large_b16      # ~10 min of CPU: sam2.1_hiera_l, 16 objects, 9 frames

TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from det_sam2_amd.config import resolve_config  # noqa: E402
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame  # noqa: E402
from det_sam2_amd.weights import synthetic_state_dict  # noqa: E402
from oracle.video_processor import OracleVideoProcessor  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# the measured shape of tests/test_hip_measured_shape.py / oracle/make_goldens.py L16_KW: one reverse pass over 9 frames
L16_KW = dict(skip_classes=set(), frame_buffer_size=9, detect_interval=9, max_frame_num_to_track=9, max_inference_state_frames=-1)
L16_FRAMES = 9


def fill8_large_b16(name="sam2.1_hiera_l", objects=16, fname="oracle_fill8_large_b16.npz"):
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, 0)
    rows = {}
    for area in (8, 0):          # 0: the same run without filling, to state how many pixels the filling moved
        vp = OracleVideoProcessor(sd, cfg, SyntheticDetector(objects), fill_hole_area=area, **L16_KW)
        t0 = time.time()
        with torch.inference_mode():
            for t in range(L16_FRAMES):
                vp.process_frame(t, synthetic_frame(t))
        dt = time.time() - t0
        od = vp.inference_state["output_dict"]
        rows[area] = (vp, od, dt)
    vp, od, dt = rows[8]
    frames = vp.pass_log[0][1]
    out = {"seconds": np.float64(dt), "frames": np.array(frames), "nobj": np.int64(objects)}
    moved = 0
    for i, t in enumerate(frames):
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        low = od[key][t]["pred_masks"].numpy()                                   # [B,1,256,256] after hole filling
        low0 = rows[0][1][key][t]["pred_masks"].numpy()
        moved += int(((low > 0) != (low0 > 0)).sum())
        seg = np.stack([vp.video_segments[t][o] for o in range(objects)])        # [B,1,1024,1024] bool
        out[f"lowbits{i}"] = np.packbits(low > 0)
        out[f"low{i}"] = low[:, :, ::4, ::4].astype(np.float16)
        out[f"bitsfull{i}"] = np.packbits(seg)
    out["filled_lowres_pixels"] = np.int64(moved)
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print(fname, dt, "s frames", frames, "low-res pixels moved by the filling:", moved)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("DS2_GOLDEN_THREADS", "8")))
    for w in sys.argv[1:] or ["fill8_large_b16"]:
        globals()[w]()
