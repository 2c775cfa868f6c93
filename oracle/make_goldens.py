"""Generate tests/golden/* by running the REFERENCE itself (imported from /root/reference in
the build container through oracle/_ref_shims.py).  The reference cannot travel to the GPU
box, so its outputs are committed as small fixtures together with this script.

    python -m oracle.make_goldens            # the sam2.1_hiera_t fixtures (a few minutes of CPU)
    python -m oracle.make_goldens l1:sam2.1_hiera_l e2e_large e2e_b16 e2e_b17 e2e_classes e2e_mask
                                             # fixtures of the larger configs / batch sizes (tens of minutes)

Fixtures are data only: seeds, boxes, and reference outputs (sub-sampled where large).
Inputs are regenerated from seeds (det_sam2_amd.synth / det_sam2_amd.weights), never stored.
"""
from __future__ import annotations

import json
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
from oracle import _ref_shims as RS  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def l1_inputs(seed=1234):
    """Seeded module-level inputs shared by the golden generator and the tests."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    B, Nk = 2, 4096 * 2 + 8
    return dict(
        img=r(1, 3, 1024, 1024), curr=r(4096, B, 256), curr_pos=r(4096, B, 256), mem=r(Nk, B, 64),
        mem_pos=r(Nk, B, 64), pix=r(B, 256, 64, 64), masks=r(B, 1, 1024, 1024),
        coords=torch.rand(B, 2, 2, generator=g) * 1024, labels=torch.tensor([[2, 3]] * B, dtype=torch.int32),
        mask_prompt=r(B, 1, 256, 256), emb=r(B, 256, 64, 64), hr0=r(B, 32, 256, 256), hr1=r(B, 64, 128, 128),
    )


def schema():
    for name in ("sam2.1_hiera_t", "sam2.1_hiera_s", "sam2.1_hiera_b+", "sam2.1_hiera_l"):
        m = RS.instantiate_from_yaml(f"configs/sam2.1/{name}.yaml")
        sd = {k: list(v.shape) for k, v in m.state_dict().items()}
        with open(os.path.join(GOLD, f"schema_{name}.json"), "w") as f:
            json.dump(sd, f, indent=0)
        print("schema", name, len(sd))


def l1(name="sam2.1_hiera_t"):
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, 0)
    ref = RS.instantiate_from_yaml(f"configs/sam2.1/{name}.yaml", sd)
    x = l1_inputs()
    out = {}
    with torch.inference_mode():
        bo = ref.forward_image(x["img"])
        for i, f in enumerate(bo["backbone_fpn"]):
            out[f"fpn{i}"] = f[0, ::4, ::8, ::8].numpy()
        out["pos2"] = bo["vision_pos_enc"][2][0, ::8, ::8, ::8].numpy()
        ma = ref.memory_attention(curr=[x["curr"]], curr_pos=[x["curr_pos"]], memory=x["mem"],
                                  memory_pos=x["mem_pos"], num_obj_ptr_tokens=8)
        out["memattn"] = ma[::32, :, ::4].numpy()
        me = ref.memory_encoder(x["pix"], x["masks"], skip_mask_sigmoid=True)
        out["memenc"] = me["vision_features"][:, ::2, ::4, ::4].numpy()
        out["memenc_pos"] = me["vision_pos_enc"][0][0, :, ::8, ::8].numpy()
        s, d = ref.sam_prompt_encoder(points=(x["coords"], x["labels"]), boxes=None, masks=x["mask_prompt"])
        out["sparse"] = s.numpy()
        out["dense"] = d[:, ::8, ::4, ::4].numpy()
        pe = ref.sam_prompt_encoder.get_dense_pe()
        out["dense_pe"] = pe[0, ::8, ::4, ::4].numpy()
        s, d = ref.sam_prompt_encoder(points=(x["coords"], x["labels"]), boxes=None, masks=None)
        for mm in (True, False):
            r = ref.sam_mask_decoder(image_embeddings=x["emb"], image_pe=pe, sparse_prompt_embeddings=s,
                                     dense_prompt_embeddings=d, multimask_output=mm, repeat_image=False,
                                     high_res_features=[x["hr0"], x["hr1"]])
            out[f"dec{int(mm)}_masks"] = r[0][:, :, ::4, ::4].numpy()
            out[f"dec{int(mm)}_iou"] = r[1].numpy()
            out[f"dec{int(mm)}_tok"] = r[2].numpy()
            out[f"dec{int(mm)}_obj"] = r[3].numpy()
        fs = ref._forward_sam_heads(backbone_features=x["emb"], point_inputs=None, mask_inputs=None,
                                    high_res_features=[x["hr0"], x["hr1"]], multimask_output=True)
        out["heads_low"] = fs[3][:, :, ::4, ::4].numpy()
        out["heads_high"] = fs[4][:, :, ::16, ::16].numpy()
        out["heads_ptr"] = fs[5].numpy()
        out["heads_obj"] = fs[6].numpy()
    np.savez_compressed(os.path.join(GOLD, f"l1_{name}.npz"), **out)
    print("l1", name, {k: v.shape for k, v in out.items()})


def _run_reference_stream(name, n_frames, detector, weight_seed=0, logit_scale=1.0, structured=False, **vp_kwargs):
    """Drive the reference VideoProcessor.process_frame (det_sam2_RT.py:421) over synthetic frames,
    capturing every propagate_in_video yield."""
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, weight_seed, logit_scale)
    vp = RS.make_reference_video_processor(f"configs/sam2.1/{name}.yaml", sd, **vp_kwargs)
    script = []
    for t in range(n_frames):
        if vp.detect_interval != -1 and t % vp.detect_interval == 0:
            script.append([(d["coordinates"], d["class"][0], d["confidence"][0]) for d in detector(t)])
    RS.ScriptedDetector.script, RS.ScriptedDetector.cursor = script, 0
    yields, passes = [], []
    orig = vp.predictor.propagate_in_video

    def capturing(state, **kw):
        ys = []
        for t, ids, logits in orig(state, **kw):
            od = state["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            low = od[key][t]["pred_masks"].clone()
            yields.append((len(passes), t, list(ids), low.numpy(), (logits > 0).numpy()))
            ys.append(t)
            yield t, ids, logits
        passes.append((kw["start_frame_idx"], ys))

    vp.predictor.propagate_in_video = capturing
    t0 = time.time()
    for t in range(n_frames):
        vp.process_frame(t, synthetic_frame(t, structured=structured))
    if vp.frame_buffer:
        vp.Detect_and_SAM2_inference(frame_idx=n_frames - 1)
    dt = time.time() - t0
    od = vp.inference_state["output_dict"]
    final_keys = (sorted(od["cond_frame_outputs"]), sorted(od["non_cond_frame_outputs"]))
    return vp, yields, passes, final_keys, dt


def e2e_cfg1(name="sam2.1_hiera_t"):
    """BASELINE config 1: tiny, 8 frames, 1 bbox on frame 0."""
    det = SyntheticDetector(1)
    kw = dict(skip_classes=set(), frame_buffer_size=8, detect_interval=8, max_frame_num_to_track=8,
              max_inference_state_frames=-1)
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 8, det, **kw)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "low": np.stack([y[3] for y in yields]),                      # [8,1,1,256,256] fp32 logits
           "bits": np.stack([np.packbits(y[4]) for y in yields])}       # video-res (1024^2) masks, packed
    np.savez_compressed(os.path.join(GOLD, "e2e_cfg1.npz"), **out)
    print("e2e_cfg1", dt, "s", out["frames"], passes, final_keys)


def e2e_dup(name="sam2.1_hiera_t"):
    """Two boxes of the SAME class on the prompted frame: the second add_new_points_or_box call feeds the first
    call's clamped logits back as a mask prompt (prev_sam_mask_logits, sam2_video_predictor.py:470-483)."""
    det = SyntheticDetector(2, duplicates={0: 1})
    kw = dict(skip_classes=set(), frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=4,
              max_inference_state_frames=-1)
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 4, det, **kw)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "low": np.stack([y[3] for y in yields]),
           "bits": np.stack([np.packbits(y[4]) for y in yields])}
    np.savez_compressed(os.path.join(GOLD, "e2e_dup.npz"), **out)
    print("e2e_dup", dt, "s", out["frames"], out["low"].shape, passes, final_keys)


PRELOAD_A = dict(skip_classes=set(), frame_buffer_size=3, detect_interval=1, max_frame_num_to_track=3,
                 max_inference_state_frames=-1)        # bank-building run: every frame is a conditioning frame
PRELOAD_B = dict(skip_classes=set(), frame_buffer_size=4, detect_interval=-1, max_frame_num_to_track=4,
                 max_inference_state_frames=-1)        # preloaded run: no detector, tracks from the bank alone


def e2e_preload(name="sam2.1_hiera_t"):
    """A18: run A builds a 3-frame bank (2 objects, a box on every frame) and pickles it with the reference's own
    save_inference_state (det_sam2_RT.py:489-497); run B loads it through the prologue of VideoProcessor.run
    (:539-549: load, mark preload cond frames, pre_frames, init_preloading_state) and tracks 4 new frames with
    detect_interval=-1."""
    det = SyntheticDetector(2)
    vpa, ya, pa, ka, dta = _run_reference_stream(name, 3, det, **PRELOAD_A)
    scratch = os.path.join(os.path.dirname(GOLD), "_scratch")      # inside the repo, removed below
    os.makedirs(scratch, exist_ok=True)
    tmp = os.path.join(scratch, "bank.pkl")
    vpa.save_inference_state(tmp)
    cfg = resolve_config(name)
    vp = RS.make_reference_video_processor(f"configs/sam2.1/{name}.yaml", synthetic_state_dict(cfg, 0), **PRELOAD_B)
    vp.inference_state = vp.load_inference_state(tmp)
    od = vp.inference_state["output_dict"]
    vp.inference_state["preloading_memory_cond_frame_idx"] = list(od["cond_frame_outputs"].keys())
    vp.inference_state["preloading_memory_non_cond_frames_idx"] = list(od["non_cond_frame_outputs"].keys())
    vp.pre_frames = vp.inference_state["num_frames"]
    vp.predictor.init_preloading_state(vp.inference_state)
    yields = []
    orig = vp.predictor.propagate_in_video

    def capturing(state, **kw):
        for t, ids, logits in orig(state, **kw):
            o = state["output_dict"]
            key = "cond_frame_outputs" if t in o["cond_frame_outputs"] else "non_cond_frame_outputs"
            yields.append((t, list(ids), o[key][t]["pred_masks"].clone().numpy(), (logits > 0).numpy()))
            yield t, ids, logits

    vp.predictor.propagate_in_video = capturing
    t0 = time.time()
    for i in range(4):
        vp.process_frame(vp.pre_frames + i, synthetic_frame(100 + i))
    dt = time.time() - t0
    out = {"seconds": np.float64(dt), "frames": np.array([y[0] for y in yields]),
           "low": np.stack([y[2] for y in yields]), "bits": np.stack([np.packbits(y[3]) for y in yields]),
           "bank_low": np.stack([y[3] for y in ya])}
    np.savez_compressed(os.path.join(GOLD, "e2e_preload.npz"), **out)
    os.remove(tmp)
    os.rmdir(scratch)
    print("e2e_preload", dta, dt, "s", out["frames"], out["low"].shape, sorted(vp.video_segments))


def e2e_stream2(name="sam2.1_hiera_t"):
    """Two-pass stream exercising second-visit tracking, release_old_frames and the online
    new-object path (A17): 8 frames, buffer 4, detect every 4, track 8, keep 6; objects 0,1 from
    frame 0 and object 2 first detected on frame 4."""
    det = SyntheticDetector(3, appear={2: 4})
    kw = dict(skip_classes=set(), frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=8,
              max_inference_state_frames=6)
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 8, det, **kw)
    out = {"seconds": np.float64(dt), "pass_id": np.array([y[0] for y in yields]),
           "frames": np.array([y[1] for y in yields]), "nobj": np.array([len(y[2]) for y in yields]),
           "final_cond": np.array(final_keys[0]), "final_noncond": np.array(final_keys[1]),
           "images_idx": np.array(vp.inference_state["images_idx"])}
    for i, y in enumerate(yields):
        out[f"low{i}"] = y[3].astype(np.float16)                              # [B,1,256,256] logits (fp16)
        out[f"lowbits{i}"] = np.packbits(y[3] > 0)                            # exact sign of low-res logits
        out[f"bits{i}"] = np.packbits(y[4][:, :, ::2, ::2])                   # video-res masks, 2x decimated
    np.savez_compressed(os.path.join(GOLD, "e2e_stream2.npz"), **out)
    print("e2e_stream2", dt, "s", passes, final_keys, vp.inference_state["images_idx"])


def _compact(out, i, low, hi_bits):
    """Compact per-yield record for many-object fixtures: exact sign of every low-res logit (packed), the logits
    themselves 4x decimated in fp16, and the video-res masks 4x decimated (packed)."""
    out[f"lowbits{i}"] = np.packbits(low > 0)
    out[f"low{i}"] = low[:, :, ::4, ::4].astype(np.float16)
    out[f"bits{i}"] = np.packbits(hi_bits[:, :, ::4, ::4])


LARGE_KW = dict(skip_classes=set(), frame_buffer_size=3, detect_interval=3, max_frame_num_to_track=3,
                max_inference_state_frames=-1)


def e2e_large(name="sam2.1_hiera_l"):
    """The headline model end to end: sam2.1_hiera_l, 3 frames, 2 objects, one pass (the scenario of
    tests/test_hip_e2e.py::test_hiera_large_matches_reference)."""
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 3, SyntheticDetector(2), **LARGE_KW)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "low": np.stack([y[3] for y in yields]),                      # [3,2,1,256,256] fp32 logits
           "bits": np.stack([np.packbits(y[4]) for y in yields])}
    np.savez_compressed(os.path.join(GOLD, "e2e_large.npz"), **out)
    print("e2e_large", dt, "s", out["frames"], out["low"].shape, passes, final_keys)


B16_KW = dict(skip_classes=set(), frame_buffer_size=3, detect_interval=3, max_frame_num_to_track=3,
              max_inference_state_frames=-1)


def e2e_b16(name="sam2.1_hiera_t"):
    """16 objects (the batch size of BASELINE configs 3-5) end to end on the tiny model: 3 frames, one pass."""
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 3, SyntheticDetector(16), **B16_KW)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "nobj": np.array([len(y[2]) for y in yields])}
    for i, y in enumerate(yields):
        _compact(out, i, y[3], y[4])
    np.savez_compressed(os.path.join(GOLD, "e2e_b16.npz"), **out)
    print("e2e_b16", dt, "s", out["frames"], passes, final_keys)


B17_KW = dict(skip_classes=set(), frame_buffer_size=2, detect_interval=2, max_frame_num_to_track=4,
              max_inference_state_frames=-1)


def e2e_b17(name="sam2.1_hiera_t"):
    """BASELINE config 5's mid-stream new category at full batch: 16 objects from frame 0, a 17th class first
    detected on frame 2 (second pass) => A17 re-consolidation of the cond frames from B=16 to B=17 and a reverse
    pass over 4 frames with 17 objects."""
    det = SyntheticDetector(17, appear={16: 2})
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 4, det, **B17_KW)
    out = {"seconds": np.float64(dt), "pass_id": np.array([y[0] for y in yields]),
           "frames": np.array([y[1] for y in yields]), "nobj": np.array([len(y[2]) for y in yields]),
           "final_cond": np.array(final_keys[0]), "final_noncond": np.array(final_keys[1])}
    for i, y in enumerate(yields):
        _compact(out, i, y[3], y[4])
    np.savez_compressed(os.path.join(GOLD, "e2e_b17.npz"), **out)
    print("e2e_b17", dt, "s", out["frames"], out["nobj"], passes, final_keys)


CLASSES_KW = dict(frame_buffer_size=2, detect_interval=1, max_frame_num_to_track=2, max_inference_state_frames=-1)
CLASSES_IDS = [3, 11, 14, 7, 11]       # detector objects -> YOLO class ids: 11 = special (collected, not tracked), 14 skipped


def e2e_classes(name="sam2.1_hiera_t"):
    """A2 branches with the reference's DEFAULT skip_classes {11, 14, 15, 19} and special class 11
    (det_sam2_RT.py:201-265,267-316): classes 3 and 7 are tracked, 14 is skipped, the two class-11 boxes are collected
    in special_classes_detection (first on frame 0 one box - object 4 appears on frame 1, then two)."""
    det = SyntheticDetector(5, class_ids=CLASSES_IDS, appear={4: 1})
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 2, det, **CLASSES_KW)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "obj_ids": np.array(yields[0][2]),
           "special": np.stack([np.asarray(b, dtype=np.float32).reshape(-1) for b in vp.special_classes_detection]),
           "special_count": np.int64(vp.special_classes_count),
           "low": np.stack([y[3] for y in yields]),
           "bits": np.stack([np.packbits(y[4]) for y in yields])}
    np.savez_compressed(os.path.join(GOLD, "e2e_classes.npz"), **out)
    print("e2e_classes", dt, "s", out["frames"], out["obj_ids"], out["special"], out["low"].shape)


def mask_prompts():
    """Seeded mask prompts of e2e_mask: a disc at model resolution (no resize) and an ellipse given at 512x384
    (exercises the antialiased resize + >= 0.5 branch, sam2_video_predictor.py:552-561)."""
    yy, xx = np.mgrid[0:1024, 0:1024]
    m0 = ((xx - 300) ** 2 + (yy - 420) ** 2) < 150 ** 2
    yy, xx = np.mgrid[0:384, 0:512]
    m1 = (((xx - 350) / 90.0) ** 2 + ((yy - 200) / 60.0) ** 2) < 1.0
    return m0, m1


def e2e_mask(name="sam2.1_hiera_t"):
    """F3: add_new_mask / _use_mask_as_output (sam2_video_predictor.py:527-616, sam2_base.py:399-448) through the
    reference predictor: two mask-prompted objects on frame 0 (+ an all-empty mask for a third), forward propagation
    over 4 frames."""
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, 0)
    ref = RS.instantiate_from_yaml(f"configs/sam2.1/{name}.yaml", sd)
    frames = [synthetic_frame(t) for t in range(4)]
    m0, m1 = mask_prompts()
    t0 = time.time()
    with torch.inference_mode():
        st = ref.init_state(frames, offload_video_to_cpu=True, offload_state_to_cpu=False)
        prompt_out = []
        for oid, m in ((0, m0), (1, m1), (2, np.zeros((1024, 1024), bool))):
            _, ids, vr = ref.add_new_mask(st, 0, oid, m)
            prompt_out.append(np.packbits((vr > 0).numpy()))
        yields = []
        for t, ids, logits in ref.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=4, reverse=False):
            od = st["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            yields.append((t, list(ids), od[key][t]["pred_masks"].clone().numpy(), (logits > 0).numpy(),
                           od[key][t]["obj_ptr"].clone().numpy(), od[key][t]["object_score_logits"].clone().numpy()))
    dt = time.time() - t0
    out = {"seconds": np.float64(dt), "frames": np.array([y[0] for y in yields]),
           "low": np.stack([y[2] for y in yields]), "bits": np.stack([np.packbits(y[3]) for y in yields]),
           "obj_ptr0": yields[0][4], "obj_score0": yields[0][5],
           "prompt_bits2": prompt_out[2]}      # video-res masks returned by the third add_new_mask call (all 3 objects)
    np.savez_compressed(os.path.join(GOLD, "e2e_mask.npz"), **out)
    print("e2e_mask", dt, "s", out["frames"], out["low"].shape, out["obj_score0"].ravel())


def correction_prompts(size=1024):
    """Seeded correction prompts of e2e_correct (points in video pixels)."""
    from det_sam2_amd.synth import synthetic_box
    b0, b1 = synthetic_box(0, 0, size=size), synthetic_box(1, 0, size=size)
    c0 = np.array([[(b0[0] + b0[2]) / 2, (b0[1] + b0[3]) / 2]], np.float32)
    c1 = np.array([[(b1[0] + b1[2]) / 2, (b1[1] + b1[3]) / 2], [b1[0] + 12.0, b1[1] + 15.0]], np.float32)
    yy, xx = np.mgrid[0:size, 0:size]
    disc = ((xx - 0.375 * size) ** 2 + (yy - 0.125 * size) ** 2) < (0.06 * size) ** 2
    return [("points", 3, 0, c0, np.array([0], np.int32)),            # a negative click on object 0, frame 3 (1 point)
            ("points", 4, 1, c1, np.array([1, 0], np.int32)),         # a positive + a negative click on object 1, frame 4
            ("mask", 2, 1, disc, None)]                               # a mask for object 1 on frame 2


def e2e_correct(name="sam2.1_hiera_t"):
    """A6 completeness: correction prompts on ALREADY-TRACKED frames (sam2_video_predictor.py:428-483,583-586; preflight
    :836-857) through the reference predictor: boxes for 2 objects on frame 0, forward propagation over 6 frames, then a
    negative click (frame 3, object 0), two clicks (frame 4, object 1) and a mask (frame 2, object 1), then propagation
    again - frames 2, 3, 4 come back corrected (consolidated non-conditioning outputs), frames 1 and 5 are re-tracked."""
    from det_sam2_amd.synth import synthetic_box
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, 0)
    ref = RS.instantiate_from_yaml(f"configs/sam2.1/{name}.yaml", sd)
    # The Det-SAM2 copy of SAM2VideoPredictor.__init__ (sam2_video_predictor.py:24-40) lost upstream's
    # `add_all_frames_to_correct_as_cond` argument while :463 / :581 still read it, so AS SHIPPED a prompt on a tracked
    # frame dies with AttributeError.  The harness sets the attribute to upstream's default (False) - no reference file is
    # edited - so that the path the attribute guards can be pinned.
    assert not hasattr(ref, "add_all_frames_to_correct_as_cond")
    ref.add_all_frames_to_correct_as_cond = False
    frames = [synthetic_frame(t) for t in range(6)]
    t0 = time.time()
    out = {}
    with torch.inference_mode():
        st = ref.init_state(frames, offload_video_to_cpu=True, offload_state_to_cpu=False)
        for o in range(2):
            ref.add_new_points_or_box(st, 0, o, box=synthetic_box(o, 0))
        first = [(t, (lg > 0).numpy()) for t, ids, lg in ref.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6)]
        out["first_frames"] = np.array([t for t, _ in first])
        out["first_bits"] = np.stack([np.packbits(b) for _, b in first])
        for i, (kind, t, oid, a, b) in enumerate(correction_prompts()):
            if kind == "points":
                _, ids, vr = ref.add_new_points_or_box(st, t, oid, points=a, labels=b)
            else:
                _, ids, vr = ref.add_new_mask(st, t, oid, a)
            out[f"prompt_bits{i}"] = np.packbits((vr > 0).numpy())
            tmp = st["temp_output_dict_per_obj"][oid]
            assert t in tmp["non_cond_frame_outputs"] and t not in tmp["cond_frame_outputs"]
            out[f"prompt_low{i}"] = tmp["non_cond_frame_outputs"][t]["pred_masks"].clone().numpy()
        yields = []
        for t, ids, logits in ref.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6):
            od = st["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            yields.append((t, od[key][t]["pred_masks"].clone().numpy(), (logits > 0).numpy()))
    dt = time.time() - t0
    od = st["output_dict"]
    out.update(seconds=np.float64(dt), frames=np.array([y[0] for y in yields]), low=np.stack([y[1] for y in yields]),
               bits=np.stack([np.packbits(y[2]) for y in yields]),
               final_cond=np.array(sorted(od["cond_frame_outputs"])), final_noncond=np.array(sorted(od["non_cond_frame_outputs"])),
               consolidated_noncond=np.array(sorted(st["consolidated_frame_inds"]["non_cond_frame_outputs"])))
    np.savez_compressed(os.path.join(GOLD, "e2e_correct.npz"), **out)
    print("e2e_correct", dt, "s", out["frames"], out["low"].shape, out["final_cond"], out["final_noncond"],
          out["consolidated_noncond"], "changed px by corrections:",
          [int((np.unpackbits(out[f"prompt_bits{i}"]) != np.unpackbits(out["first_bits"][t].reshape(-1))).sum())
           for i, (_, t, _, _, _) in enumerate(correction_prompts())])


def removal_click(size=1024):
    """The positive click of e2e_remove: the centre of object 1's box on frame 3 (video pixels)."""
    from det_sam2_amd.synth import synthetic_box
    b = synthetic_box(1, 3, size=size)
    return np.array([[(b[0] + b[2]) / 2, (b[1] + b[3]) / 2]], np.float32), np.array([1], np.int32)


def e2e_remove(name="sam2.1_hiera_t"):
    """Prompt / object removal (sam2_video_predictor.py:1061-1131 clear_all_prompts_in_frame, :1438-1549 remove_object)
    through the reference predictor: boxes for 3 objects on frame 0 and a click for object 1 on the untracked frame 3 (a
    second conditioning frame that only object 1 has an input on), forward propagation over 6 frames; a correction click
    on object 0, frame 4 that is cleared again (clear_all_prompts_in_frame); then remove_object(1) - frame 3 is demoted
    to a non-conditioning frame, every stored entry loses row 1 - and propagation again with the 2 remaining objects."""
    from det_sam2_amd.synth import synthetic_box
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, 0)
    ref = RS.instantiate_from_yaml(f"configs/sam2.1/{name}.yaml", sd)
    ref.add_all_frames_to_correct_as_cond = False          # see e2e_correct
    frames = [synthetic_frame(t) for t in range(6)]
    pts, lab = removal_click()
    t0 = time.time()
    out = {}
    with torch.inference_mode():
        st = ref.init_state(frames, offload_video_to_cpu=True, offload_state_to_cpu=False)
        for o in range(3):
            ref.add_new_points_or_box(st, 0, o, box=synthetic_box(o, 0))
        ref.add_new_points_or_box(st, 3, 1, points=pts, labels=lab)
        first = [(t, (lg > 0).numpy()) for t, ids, lg in ref.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6)]
        out["first_frames"] = np.array([t for t, _ in first])
        out["first_bits"] = np.stack([np.packbits(b) for _, b in first])
        out["first_cond"] = np.array(sorted(st["output_dict"]["cond_frame_outputs"]))
        # a correction on a tracked frame, then cleared again: the frame shows the tracked masks
        c0 = correction_prompts()[0]
        _, _, vr = ref.add_new_points_or_box(st, 4, 0, points=c0[3], labels=c0[4])
        out["click_bits"] = np.packbits((vr > 0).numpy())
        t, ids, vr = ref.clear_all_prompts_in_frame(st, 4, 0)
        out["clear_bits"] = np.packbits((vr > 0).numpy())
        assert t == 4 and list(ids) == [0, 1, 2]
        assert 4 not in st["temp_output_dict_per_obj"][0]["non_cond_frame_outputs"]
        ids, updated = ref.remove_object(st, 1)
        out["ids_after"] = np.array(list(ids))
        out["updated_frames"] = np.array([t for t, _ in updated])
        out["updated_bits"] = np.stack([np.packbits((m > 0).numpy()) for _, m in updated])
        od = st["output_dict"]
        out["cond_after"] = np.array(sorted(od["cond_frame_outputs"]))
        out["noncond_after"] = np.array(sorted(od["non_cond_frame_outputs"]))
        out["tracked_after"] = np.array(sorted(st["frames_already_tracked"]))
        out["consolidated_cond_after"] = np.array(sorted(st["consolidated_frame_inds"]["cond_frame_outputs"]))
        out["low3_after"] = od["non_cond_frame_outputs"][3]["pred_masks"].clone().numpy()
        ids2, upd2 = ref.remove_object(st, 77)                      # unknown id, strict=False: no-op
        assert list(ids2) == list(ids) and upd2 == []
        yields = []
        for t, ids, logits in ref.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6):
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            yields.append((t, od[key][t]["pred_masks"].clone().numpy(), (logits > 0).numpy()))
    dt = time.time() - t0
    out.update(seconds=np.float64(dt), frames=np.array([y[0] for y in yields]), low=np.stack([y[1] for y in yields]),
               bits=np.stack([np.packbits(y[2]) for y in yields]))
    np.savez_compressed(os.path.join(GOLD, "e2e_remove.npz"), **out)
    print("e2e_remove", dt, "s", out["frames"], out["low"].shape, "ids", out["ids_after"], "updated", out["updated_frames"],
          "cond", out["cond_after"], "noncond", out["noncond_after"], "tracked", out["tracked_after"])


# ---------------------------------------------------------------------------------------------- held-out family
# the arithmetic mode bf16x3k was selected against the fixtures above (weight seed 0, uniform-noise
# frames, |logit| up to 14-17).  The fixtures below were generated AFTER that choice and differ in all three respects:
# weight seed 1, structured frames (moving discs on a gradient, synth.synthetic_frame(structured=True)), and - the "lm"
# variants - the hypernetwork output layer scaled by 1/30 so that |logit| < 1 (non-saturated sigmoids into the memory
# encoder, objectness / best-of-3 selection at small margins: sam2_base.py:343-350,363-370).
HELDOUT = {            # variant -> (weight_seed, logit_scale, structured frames)
    "s1": (1, 1.0, True),
    "lm": (1, 1.0 / 30.0, True),
    "s2": (2, 1.0, True),              # round 5: a second structured-frame seed for the fixtures at the measured shape / config 3's model
}


def heldout_cfg1(variant="s1", name="sam2.1_hiera_t"):
    """BASELINE config 1 (tiny, 8 frames, 1 box) under a held-out variant."""
    ws, ls, st = HELDOUT[variant]
    kw = dict(skip_classes=set(), frame_buffer_size=8, detect_interval=8, max_frame_num_to_track=8,
              max_inference_state_frames=-1)
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 8, SyntheticDetector(1), ws, ls, st, **kw)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "low": np.stack([y[3] for y in yields]), "bits": np.stack([np.packbits(y[4]) for y in yields])}
    np.savez_compressed(os.path.join(GOLD, f"ho_cfg1_{variant}.npz"), **out)
    print("heldout_cfg1", variant, dt, "s", out["frames"], "logit absmax", float(np.abs(out["low"]).max()),
          "fg fraction", float((out["low"] > 0).mean()))


def heldout_b16(variant="s1", name="sam2.1_hiera_t"):
    """16 objects, 3 frames, one pass (e2e_b16's scenario) under a held-out variant."""
    ws, ls, st = HELDOUT[variant]
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 3, SyntheticDetector(16), ws, ls, st, **B16_KW)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "nobj": np.array([len(y[2]) for y in yields]),
           "logit_absmax": np.float32(max(float(np.abs(y[3]).max()) for y in yields))}
    for i, y in enumerate(yields):
        _compact(out, i, y[3], y[4])
    np.savez_compressed(os.path.join(GOLD, f"ho_b16_{variant}.npz"), **out)
    print("heldout_b16", variant, dt, "s", out["frames"], "logit absmax", float(out["logit_absmax"]))


def heldout_large(variant="s1", name="sam2.1_hiera_l"):
    """sam2.1_hiera_l, 3 frames, 2 objects (e2e_large's scenario) under a held-out variant."""
    ws, ls, st = HELDOUT[variant]
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, 3, SyntheticDetector(2), ws, ls, st, **LARGE_KW)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "low": np.stack([y[3] for y in yields]), "bits": np.stack([np.packbits(y[4]) for y in yields])}
    np.savez_compressed(os.path.join(GOLD, f"ho_large_{variant}.npz"), **out)
    print("heldout_large", variant, dt, "s", out["frames"], "logit absmax", float(np.abs(out["low"]).max()),
          "fg fraction", float((out["low"] > 0).mean()))


# ---------------------------------------------------------------------------------------------- round 4: measured shape
# VERDICT r3 missing #2: an end-to-end reference fixture AT the benchmark's shape (sam2.1_hiera_l x 16 objects with the
# bank grown to 1 conditioning + 6 non-conditioning frames, Nk = 28 736) and one for BASELINE config 3's model
# (sam2.1_hiera_base_plus, preloaded bank).
L16_KW = dict(skip_classes=set(), frame_buffer_size=9, detect_interval=9, max_frame_num_to_track=9,
              max_inference_state_frames=-1)
L16_FRAMES = 9


def _large_b16(fname, weight_seed=0, logit_scale=1.0, structured=False, name="sam2.1_hiera_l"):
    vp, yields, passes, final_keys, dt = _run_reference_stream(name, L16_FRAMES, SyntheticDetector(16), weight_seed,
                                                               logit_scale, structured, **L16_KW)
    out = {"seconds": np.float64(dt), "frames": np.array([y[1] for y in yields]),
           "nobj": np.array([len(y[2]) for y in yields]),
           "logit_absmax": np.float32(max(float(np.abs(y[3]).max()) for y in yields))}
    for i, y in enumerate(yields):
        _compact(out, i, y[3], y[4])
        # the video-resolution masks in FULL (packed; blobs compress well): BASELINE's bar is the IoU of THESE masks, and on
        # masks of ~5 500 low-res pixels (the structured frames) a 4x decimated copy turns one boundary pixel into 2e-4
        out[f"bitsfull{i}"] = np.packbits(y[4])
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print(fname, dt, "s", out["frames"], passes, final_keys, "logit absmax", float(out["logit_absmax"]))


def e2e_large_b16():
    """The BENCHMARK's shape end to end through the reference: sam2.1_hiera_l, 16 objects, ONE reverse pass over 9 frames
    with the boxes on frame 0 - frame 8 is tracked against the conditioning frame alone, frame 7 against 1 + 1, ...,
    frames 2 and 1 against 1 conditioning + 6 non-conditioning frames + 4 x (1 + 6 ... 7) pointer tokens
    (sam2_video_predictor.py:911-1025 driven by det_sam2_RT.py:342-411; bank sam2_base.py:479-690)."""
    _large_b16("e2e_large_b16.npz")


def heldout_large_b16(variant="s1"):
    ws, ls, st = HELDOUT[variant]
    _large_b16(f"ho_large_b16_{variant}.npz", ws, ls, st)


BPLUS_A = dict(skip_classes=set(), frame_buffer_size=1, detect_interval=1, max_frame_num_to_track=1,
               max_inference_state_frames=-1)        # bank-building run: ONE conditioning frame (P = 1)
BPLUS_B = dict(skip_classes=set(), frame_buffer_size=4, detect_interval=-1, max_frame_num_to_track=4,
               max_inference_state_frames=-1)        # preloaded run: no detector, 4 frames tracked from the bank
BPLUS_OBJECTS = 4


def _bplus(fname, weight_seed=0, logit_scale=1.0, structured=False, name="sam2.1_hiera_b+"):
    det = SyntheticDetector(BPLUS_OBJECTS)
    vpa, ya, pa, ka, dta = _run_reference_stream(name, 1, det, weight_seed, logit_scale, structured, **BPLUS_A)
    scratch = os.path.join(os.path.dirname(GOLD), "_scratch")
    os.makedirs(scratch, exist_ok=True)
    tmp = os.path.join(scratch, "bank_bplus.pkl")
    vpa.save_inference_state(tmp)
    cfg = resolve_config(name)
    vp = RS.make_reference_video_processor(f"configs/sam2.1/{name}.yaml", synthetic_state_dict(cfg, weight_seed, logit_scale),
                                           **BPLUS_B)
    vp.inference_state = vp.load_inference_state(tmp)
    od = vp.inference_state["output_dict"]
    vp.inference_state["preloading_memory_cond_frame_idx"] = list(od["cond_frame_outputs"].keys())
    vp.inference_state["preloading_memory_non_cond_frames_idx"] = list(od["non_cond_frame_outputs"].keys())
    vp.pre_frames = vp.inference_state["num_frames"]
    vp.predictor.init_preloading_state(vp.inference_state)
    yields = []
    orig = vp.predictor.propagate_in_video

    def capturing(state, **kw):
        for t, ids, logits in orig(state, **kw):
            o = state["output_dict"]
            key = "cond_frame_outputs" if t in o["cond_frame_outputs"] else "non_cond_frame_outputs"
            yields.append((t, list(ids), o[key][t]["pred_masks"].clone().numpy(), (logits > 0).numpy()))
            yield t, ids, logits

    vp.predictor.propagate_in_video = capturing
    t0 = time.time()
    for i in range(4):
        vp.process_frame(vp.pre_frames + i, synthetic_frame(100 + i, structured=structured))
    dt = time.time() - t0
    out = {"seconds": np.float64(dt), "frames": np.array([y[0] for y in yields]),
           "low": np.stack([y[2] for y in yields]).astype(np.float32),            # [4,4,1,256,256] fp32 logits
           "bits": np.stack([np.packbits(y[3]) for y in yields])}
    assert not ya          # a reverse pass starting on frame 0 yields nothing: the bank holds the consolidated cond frame only
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    os.remove(tmp)
    os.rmdir(scratch)
    print(fname, dta, dt, "s", out["frames"], out["low"].shape, sorted(vp.video_segments),
          "logit absmax", float(np.abs(out["low"]).max()))


def e2e_bplus():
    """BASELINE config 3's model end to end through the reference: sam2.1_hiera_base_plus, 4 objects; run A prompts one
    frame and pickles the state with the reference's save_inference_state (det_sam2_RT.py:489-497) - a bank of P = 1
    conditioning frame; run B preloads it (:539-549, sam2_video_predictor.py:123-156) and tracks 4 new frames with
    detect_interval = -1 (one reverse pass: 1 preload cond frame + 0..3 non-conditioning frames)."""
    _bplus("e2e_bplus.npz")


def heldout_bplus(variant="s1"):
    ws, ls, st = HELDOUT[variant]
    _bplus(f"ho_bplus_{variant}.npz", ws, ls, st)



def e2e_nopost(name="sam2.1_hiera_t"):
    """build_sam2_video_predictor(..., apply_postprocessing=False) (build_sam.py:111-146 without the overrides of :126-135)
    through the reference predictor: single-mask decoder output = token 0 (no dynamic multimask fallback; a box prompt is two
    points > multimask_max_pt_num), prompted masks through the SIGMOID into the memory encoder, no hole filling.  Boxes for 2
    objects on frame 0, forward propagation over 4 frames."""
    from det_sam2_amd.synth import synthetic_box
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, 0)
    ref = RS.instantiate_from_yaml(f"configs/sam2.1/{name}.yaml", sd, apply_postprocessing=False)
    assert not ref.sam_mask_decoder.dynamic_multimask_via_stability and not ref.binarize_mask_from_pts_for_mem_enc
    frames = [synthetic_frame(t) for t in range(4)]
    t0 = time.time()
    with torch.inference_mode():
        st = ref.init_state(frames, offload_video_to_cpu=True, offload_state_to_cpu=False)
        for o in range(2):
            ref.add_new_points_or_box(st, 0, o, box=synthetic_box(o, 0))
        yields = []
        for t, ids, logits in ref.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=4, reverse=False):
            od = st["output_dict"]
            key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
            yields.append((t, od[key][t]["pred_masks"].clone().numpy(), (logits > 0).numpy()))
    dt = time.time() - t0
    out = {"seconds": np.float64(dt), "frames": np.array([y[0] for y in yields]), "low": np.stack([y[1] for y in yields]),
           "bits": np.stack([np.packbits(y[2]) for y in yields])}
    np.savez_compressed(os.path.join(GOLD, "e2e_nopost.npz"), **out)
    print("e2e_nopost", dt, "s", out["frames"], out["low"].shape)



PIPE_IDS = [2, 11, 6]          # detector objects -> YOLO class ids: 11 is special (collected, skipped by default), 2 and 6 are tracked
PIPE_SMALL_KW = dict(frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=8, max_inference_state_frames=2000)


class _PipelineRecorder:
    """Stands where the reference's VideoPostProcessor stands (Det_SAM2_pipeline.py:48,150-151,193-208) and records what the
    consumer thread hands it: the relative frame index of every accepted delivery with a copy of its masks."""

    def __init__(self):
        outer = self
        self.special, self.deliveries, self._pending = None, [], None

        class _Positions(dict):
            def __setitem__(s, frame_idx, token):      # balls_positions[frame_idx] = process_frame_positions(segments), :196
                outer.deliveries.append((int(frame_idx), outer._pending))
                dict.__setitem__(s, frame_idx, token)

        self.balls_positions, self.balls_velocities = _Positions(), {}

    def get_hole_name(self, special):
        self.special = [np.asarray(b, np.float32).reshape(-1).copy() for b in special]

    def get_boundary_from_holes(self):
        pass

    def process_frame_positions(self, segments):
        self._pending = {int(k): np.asarray(v).copy() for k, v in segments.items()}
        return len(self.deliveries)

    def process_frame_velocities(self, frame_idx, time_interval=1.0):
        return {}

    def check_ball_disappeared_pot(self, frame_idx):
        pass

    check_ball_collision = check_ball_rebound = check_ball_disappeared_pot


def _run_reference_pipeline(fname, n_frames, max_frames, vp_override=None, name="sam2.1_hiera_t"):
    """F1: the reference's DetSAM2Pipeline.inference (Det_SAM2_pipeline.py:81-247) itself - constructor, producer thread with
    cv2.VideoCapture loop, transform_video_segments, consumer thread - over a synthetic stream; only the third-party
    modules are harness stand-ins (oracle/_ref_shims.py) and the post-processor is a recorder.  ``vp_override`` replaces
    VideoProcessor keyword arguments that the reference constructor hard-codes (:31-46) to keep a fixture small."""
    import threading

    RS.install_shims()
    import det_sam2_RT
    import Det_SAM2_pipeline as RP          # the reference module (sys.path: /root/reference/det_sam2_inference)
    assert RP.__file__.startswith(RS.REFERENCE_ROOT), RP.__file__
    cfg = resolve_config(name)
    sd = synthetic_state_dict(cfg, 0)
    det_sam2_RT.build_sam2_video_predictor = lambda c, ckpt: RS.instantiate_from_yaml(c, sd)
    if vp_override:
        RP.VideoProcessor = lambda **kw: det_sam2_RT.VideoProcessor(**{**kw, **vp_override})
    else:
        RP.VideoProcessor = det_sam2_RT.VideoProcessor
    pipe = RP.DetSAM2Pipeline(sam2_output_frame_dir="/tmp/ref_pipe_out", sam2_checkpoint_path=None,
                              sam2_config_path=f"configs/sam2.1/{name}.yaml", detect_model_weights=None,
                              output_video_dir="/tmp/ref_pipe_out")
    vp = pipe.video_processor
    det = SyntheticDetector(len(PIPE_IDS), class_ids=PIPE_IDS)
    RS.ScriptedDetector.script = [[(d["coordinates"], d["class"][0], d["confidence"][0]) for d in det(t)]
                                  for t in range(n_frames) if t % vp.detect_interval == 0]
    RS.ScriptedDetector.cursor = 0
    RS.ScriptedCapture.sources = {"synthetic": [synthetic_frame(t) for t in range(n_frames)]}
    rec = _PipelineRecorder()
    pipe.post_processor = rec
    enq = []
    put = pipe.frames_queue.put
    pipe.frames_queue.put = lambda item: (enq.append(int(item[0])), put(item))[1]
    before = set(threading.enumerate())
    t0 = time.time()
    pipe.inference(video_source="synthetic", max_frames=max_frames)
    pipe.inference_done_event.wait()
    while not pipe.frames_queue.empty():           # the consumer drains what is left (:183-185)
        time.sleep(0.05)
    time.sleep(1.0)
    for th in set(threading.enumerate()) - before:
        th.join(timeout=5.0)                       # the reference's consumer can block in Queue.get() for ever once the
    dt = time.time() - t0                          # producer is done (:188 has no timeout); its work is complete by then
    objs = sorted(rec.deliveries[0][1])
    out = {"seconds": np.float64(dt), "n_frames": np.int64(n_frames), "max_frames": np.int64(max_frames),
           "enqueued": np.array(enq), "delivered": np.array([d[0] for d in rec.deliveries]),
           "has_processed": np.array(pipe.has_processed_frames), "obj_ids": np.array(objs),
           "special": np.stack(rec.special), "left_in_pipeline": np.array(sorted(pipe.video_segments)),
           "left_in_backbone": np.array(sorted(vp.video_segments)),
           "vp_kwargs": np.array([vp.frame_buffer_size, vp.detect_interval, vp.max_frame_num_to_track,
                                  vp.max_inference_state_frames])}
    for i, (t, seg) in enumerate(rec.deliveries):
        assert sorted(seg) == objs and all(m.shape == (1, 1024, 1024) and m.dtype == bool for m in seg.values())
        m = np.stack([seg[o] for o in objs])
        out[f"bits{i}"] = np.packbits(m[:, :, ::2, ::2])                 # video-res masks, 2x decimated
        out[f"area{i}"] = m.reshape(len(objs), -1).sum(1)                # exact pixel counts at full resolution
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print(fname, dt, "s enqueued", enq, "delivered", out["delivered"], "special", out["special"], "left", out["left_in_pipeline"])


def pipeline(which="small"):
    """``small``: buffers of 4 / track 8 over 14 frames, stream ends -> the flush path (:124-131); ``cut``: max_frames
    reached with frames still buffered (:159-162: no flush); ``default``: the reference constructor's own 30 / 30 / 60
    over 70 frames."""
    if which == "small":
        _run_reference_pipeline("pipeline_small.npz", 14, 1000, PIPE_SMALL_KW)
    elif which == "cut":
        _run_reference_pipeline("pipeline_cut.npz", 12, 10, PIPE_SMALL_KW)
    else:
        _run_reference_pipeline("pipeline_default.npz", 70, 1000, None)
    sys.stdout.flush()
    os._exit(0)          # a consumer thread blocked in Queue.get() would keep the interpreter alive


if __name__ == "__main__":
    assert RS.reference_available(), "needs /root/reference (build container only)"
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["schema", "l1", "e2e_cfg1", "e2e_stream2", "e2e_dup", "e2e_preload"]
    torch.set_num_threads(int(os.environ.get("DS2_GOLDEN_THREADS", "8")))
    for w in which:
        fn, _, arg = w.partition(":")
        globals()[fn](*([arg] if arg else []))
