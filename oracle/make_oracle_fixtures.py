"""Fixtures whose expectation comes from the ORACLE (oracle/, pinned to the reference by tests/test_oracle_golden.py), for
behaviour the reference cannot produce on a CPU: hole filling.  ``fill_holes_in_mask_scores`` (sam2/utils/misc.py:365-393)
needs the reference's CUDA extension and silently returns its input without it (:389-391), so a fixture of the SHIPPING
default (build_sam.py:134 appends ``fill_hole_area=8``; sam2_video_predictor.py:1343-1346 applies it to every inferred
frame) cannot be recorded from /root/reference in this container.  The oracle restates that path with
``oracle/cc.py`` (8-connectivity labelling, scipy.ndimage.label contract) and is timed in minutes at the measured shape, so
its outputs are committed as a fixture instead of being recomputed on the GPU box.

    python -m oracle.make_oracle_fixtures fill8_large_b16      # ~10 min of CPU: sam2.1_hiera_l, 16 objects, 9 frames
    python -m oracle.make_oracle_fixtures three_pass fill8_tiny memattn_bench
                                                               # oracle outputs the GPU suite used to recompute on the GPU box's host
                                                               # cores (3 minutes of its 8): same seeds, same comparisons

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


def three_pass(fname="oracle_three_pass.npz"):
    """tests/test_hip_e2e.py::test_three_pass_stream_matches_oracle: tiny model, 12 frames, 3 overlapping reverse passes with eviction,
    3 objects (one appearing in the second pass) - error accumulation through the memory bank."""
    cfg = resolve_config("sam2.1_hiera_t")
    sd = synthetic_state_dict(cfg, 0)
    kw = dict(skip_classes=set(), frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=8, max_inference_state_frames=8)
    ovp = OracleVideoProcessor(sd, cfg, SyntheticDetector(3, appear={2: 4}), **kw)
    t0 = time.time()
    with torch.inference_mode():
        for t in range(12):
            ovp.process_frame(t, synthetic_frame(t))
    out = {"seconds": np.float64(time.time() - t0), "passes": np.array([p[0] for p in ovp.pass_log]),
           "pass_frames": np.array([len(p[1]) for p in ovp.pass_log])}
    for i, p in enumerate(ovp.pass_log):
        out[f"pass{i}"] = np.array(p[1])
    for t in range(12):
        objs = sorted(ovp.video_segments[t])
        out[f"objs{t}"] = np.array(objs)
        out[f"bits{t}"] = np.packbits(np.stack([ovp.video_segments[t][o] for o in objs]))      # [n,1,1024,1024] bool, in full
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print(fname, float(out["seconds"]), "s passes", [p[:2] for p in ovp.pass_log])


def fill8_tiny(fname="oracle_fill8_tiny.npz"):
    """tests/test_hip_cc.py::test_predictor_with_hole_filling_matches_oracle: tiny model, 2 objects, 4 frames, fill_hole_area = 8 and,
    to count what the filling moved, 0."""
    cfg = resolve_config("sam2.1_hiera_t")
    sd = synthetic_state_dict(cfg, 0)
    kw = dict(frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=4, max_inference_state_frames=-1)
    runs = {}
    for area in (8, 0):
        ovp = OracleVideoProcessor(sd, cfg, SyntheticDetector(2), skip_classes=set(), fill_hole_area=area, **kw)
        with torch.inference_mode():
            for t in range(4):
                ovp.process_frame(t, synthetic_frame(t))
        runs[area] = ovp
    out, changed = {}, 0
    for t in range(4):
        objs = sorted(runs[8].video_segments[t])
        a = np.stack([np.asarray(runs[8].video_segments[t][o]).astype(bool) for o in objs])
        b = np.stack([np.asarray(runs[0].video_segments[t][o]).astype(bool) for o in objs])
        changed += int((a != b).sum())
        out[f"objs{t}"], out[f"bits{t}"] = np.array(objs), np.packbits(a)
    out["pixels_changed_by_filling"] = np.int64(changed)
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print(fname, "pixels changed by the filling:", changed)


MEMATTN_CASES = [(16, 7, 16), (16, 7, 13), (16, 1, 3)]      # the 16-object cases (B, NF, NP) of test_memory_attention_at_bench_size


def memattn_inputs(B, NF, NP):
    """the seeded inputs of tests/test_hip_stages.py::test_memory_attention_at_bench_size (one definition for the test and this script)"""
    g = torch.Generator().manual_seed(21)
    curr = torch.randn(4096, 256, generator=g)
    feats = [torch.randn(B, 64, 64, 64, generator=g).to(torch.bfloat16) for _ in range(NF)]
    ptrs = [torch.randn(B, 256, generator=g) for _ in range(NP)]
    return curr, feats, ptrs, [6, 5, 4, 3, 2, 1, 0][:NF], [float(i) for i in range(NP)]


def memattn_bench(fname="oracle_memattn_bench.npz"):
    """the oracle's memory attention at the measured configuration's size (16 objects, 7-frame bank + 16 / 13 pointers: 17 s of host
    time each) and with a short bank; every 64th token of the result, fp32 (the 4-object cases of the test take seconds and still run the oracle)."""
    from oracle import modeling as M
    cfg = resolve_config("sam2.1_hiera_t")          # the memory-attention weights have the same shapes in every config
    sd = synthetic_state_dict(cfg, 0)
    out = {}
    for (B, NF, NP) in MEMATTN_CASES:
        curr, feats, ptrs, tpos_rows, ptr_pos = memattn_inputs(B, NF, NP)
        pos2 = M.sine_pos_2d(64, 64, 64)
        mems, poss = [], []
        for f, r in zip(feats, tpos_rows):
            mems.append(f.float().flatten(2).permute(2, 0, 1))
            poss.append(pos2[None].expand(B, -1, -1, -1).flatten(2).permute(2, 0, 1) + sd["maskmem_tpos_enc"][r])
        op = M.linear(sd, "obj_ptr_tpos_proj", M.sine_pe_1d(torch.tensor(ptr_pos) / 15.0, 256))
        op = op.unsqueeze(1).expand(-1, B, 64).repeat_interleave(4, dim=0)
        pt = torch.stack(ptrs, 0).reshape(-1, B, 4, 64).permute(0, 2, 1, 3).flatten(0, 1)
        memory, memory_pos = torch.cat(mems + [pt], 0), torch.cat(poss + [op], 0)
        vis_pos = M.sine_pos_2d(256, 64, 64).flatten(1).T
        t0 = time.time()
        with torch.inference_mode():
            ref = M.memory_attention(sd, cfg, curr[:, None].expand(-1, B, -1), vis_pos[:, None].expand(-1, B, -1), memory, memory_pos, 4 * NP)
        out[f"ref_{B}_{NF}_{NP}"] = ref.transpose(0, 1)[:, ::64].contiguous().numpy()          # [B, 64, 256]
        out[f"absmax_{B}_{NF}_{NP}"] = np.float32(ref.abs().max())
        print("memattn", (B, NF, NP), round(time.time() - t0, 1), "s")
    np.savez_compressed(os.path.join(GOLD, fname), **out)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("DS2_GOLDEN_THREADS", "8")))
    for w in sys.argv[1:] or ["fill8_large_b16"]:
        globals()[w]()
