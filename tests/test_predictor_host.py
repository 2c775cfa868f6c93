"""Host logic of the predictor state machine over the CPU stand-in for the stages (tests/_fake_hip.py): the paths whose
control flow matters more than their arithmetic.  The arithmetic of the same paths is pinned on the GPU against reference
goldens (tests/test_hip_correct.py) and in the oracle (tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from _fake_hip import fake_predictor
from det_sam2_amd.synth import synthetic_box


def _frames(n):
    rng = np.random.default_rng(3)
    return [rng.integers(0, 256, (64, 64, 3), dtype=np.uint8) for _ in range(n)]


def _tracked(p):
    return p.stats["tracked_frames"]


def test_correction_prompts_on_tracked_frames():
    """sam2_video_predictor.py:428-483 (points), :583-586 (mask), preflight :836-857: a prompt on a tracked frame runs a
    memory-conditioned single-object pass in the frame's tracking direction, is stored as a NON-conditioning temp output,
    consolidated by the next preflight (memory re-encoded), reused by the next propagation; the frames around it are
    re-tracked."""
    p = fake_predictor()
    st = p.init_state(_frames(6))
    for o in range(2):
        p.add_new_points_or_box(st, 0, o, box=synthetic_box(o, 0, size=64))
    first = {t: m.clone() for t, _, m in p.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6)}
    assert sorted(first) == list(range(6)) and _tracked(p) == 5
    assert st["frames_already_tracked"][3] == {"reverse": False}
    ma0 = p.hip.calls["memory_attention"]
    old3 = st["output_dict"]["non_cond_frame_outputs"][3]["pred_masks"].clone()
    # (1) a negative click on object 0, frame 3
    t, ids, vr = p.add_new_points_or_box(st, 3, 0, points=np.array([[20.0, 20.0]], np.float32), labels=np.array([0], np.int32))
    assert (t, list(ids)) == (3, [0, 1]) and vr.shape == (2, 1, 64, 64)
    assert p.hip.calls["memory_attention"] == ma0 + 1                       # memory-conditioned, unlike an init-cond prompt
    tmp = st["temp_output_dict_per_obj"][0]
    assert 3 in tmp["non_cond_frame_outputs"] and 3 not in tmp["cond_frame_outputs"]
    assert tmp["non_cond_frame_outputs"][3]["maskmem_features"] is None     # memory encoder deferred to the preflight
    assert not torch.equal(tmp["non_cond_frame_outputs"][3]["pred_masks"][0], old3[0])
    # the returned masks: object 0 corrected, object 1 as tracked
    low1 = p.hip.mask_output(old3[1:2, 0], 64, 64)[0]
    assert torch.equal(vr[1], low1[0])
    # (2) a second click on the same frame feeds the first correction back as mask prompt (prev_sam_mask_logits)
    p.add_new_points_or_box(st, 3, 0, points=np.array([[24.0, 20.0]], np.float32), labels=np.array([1], np.int32))
    assert p.hip.calls["memory_attention"] == ma0 + 2
    # (3) a mask on object 1, frame 2 (tracked): stored as non-cond too, no memory read
    m = np.zeros((64, 64), bool)
    m[10:30, 12:40] = True
    p.hip.resize_aa = lambda x, h, w, threshold=0.5: (torch.nn.functional.interpolate(x[None], size=(h, w))[0] >= threshold).float()
    p.add_new_mask(st, 2, 1, m)
    assert 2 in st["temp_output_dict_per_obj"][1]["non_cond_frame_outputs"]
    assert p.hip.calls["memory_attention"] == ma0 + 2
    # second propagation: frames 2, 3 come back consolidated, 1, 4, 5 are re-tracked
    n0 = _tracked(p)
    second = {t: m.clone() for t, _, m in p.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=6)}
    assert _tracked(p) == n0 + 3
    assert sorted(st["consolidated_frame_inds"]["non_cond_frame_outputs"]) == [2, 3]
    od = st["output_dict"]
    assert sorted(od["cond_frame_outputs"]) == [0] and sorted(od["non_cond_frame_outputs"]) == [1, 2, 3, 4, 5]
    assert od["non_cond_frame_outputs"][3]["maskmem_features"] is not None
    assert all(not d["non_cond_frame_outputs"] and not d["cond_frame_outputs"] for d in st["temp_output_dict_per_obj"].values())
    assert torch.equal(second[0], first[0]) and torch.equal(second[1], first[1])      # before the corrections: unchanged
    assert not torch.equal(second[3], first[3]) and not torch.equal(second[4], first[4])   # corrected / downstream of it


def test_prompt_on_unencodable_frame_is_a_clear_error():
    p = fake_predictor()
    st = p.init_state(_frames(4))
    p.add_new_points_or_box(st, 0, 0, box=synthetic_box(0, 0, size=64))
    list(p.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=4))
    p.release_old_frames(st, 3, 2, 0, release_images=True)                 # frames 0, 1 are gone
    with pytest.raises(RuntimeError, match="cannot be encoded"):
        p.add_new_points_or_box(st, 1, 0, points=np.array([[5.0, 5.0]], np.float32), labels=np.array([1], np.int32))


def test_run_from_a_frame_folder_equals_run_from_memory(tmp_path):
    """VideoProcessor.run(frame_dir=...) (det_sam2_RT.py:580-598; load_frames_from_folder :507-524): PNG frames decoded
    from a folder in sorted name order give exactly the masks of the same frames handed over in memory."""
    from PIL import Image
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.synth import SyntheticDetector
    frames = _frames(7)
    for i, f in enumerate(frames):
        Image.fromarray(f).save(tmp_path / f"{i:05d}.png")
    (tmp_path / "notes.txt").write_text("not a frame")
    kw = dict(model_cfg="sam2.1_hiera_t", skip_classes=set(), frame_buffer_size=3, detect_interval=3, max_frame_num_to_track=6,
              max_inference_state_frames=6)
    a = VideoProcessor(detector=SyntheticDetector(2, size=64), predictor=fake_predictor(), **kw)
    segs_a = a.run(frames=frames)
    b = VideoProcessor(detector=SyntheticDetector(2, size=64), predictor=fake_predictor(), **kw)
    segs_b = b.run(frame_dir=str(tmp_path))
    assert sorted(segs_a) == sorted(segs_b) == list(range(7))
    for t in segs_a:
        for o in segs_a[t]:
            assert np.array_equal(segs_a[t][o], segs_b[t][o])
    lst = b.load_frames_from_folder(str(tmp_path))          # list-like, as the reference's return value (:507-524)
    assert len(lst) == 7 and np.array_equal(lst[2], frames[2]) and np.array_equal(lst[-1], frames[-1]) and len(lst[1:3]) == 2
    empty = tmp_path / "empty"
    empty.mkdir()
    assert VideoProcessor(detector=SyntheticDetector(2, size=64), predictor=fake_predictor(), **kw).run(frame_dir=str(empty)) is None
    (empty / "00000.png").write_bytes(b"not a png")        # only unreadable files: returns early like the reference (:586-588)
    assert VideoProcessor(detector=SyntheticDetector(2, size=64), predictor=fake_predictor(), **kw).run(frame_dir=str(empty)) is None
    with pytest.raises(NotImplementedError, match="OpenCV"):
        VideoProcessor(detector=SyntheticDetector(2, size=64), predictor=fake_predictor(), **kw).run(video_path="x.mp4")
    # a truncated file is skipped - at construction when the header check notices (PNG: chunk CRCs, no pixel decoded), else
    # when the stream reaches it (JPEG) - and the remaining frames run as if it were not there
    raw = (tmp_path / "00003.png").read_bytes()
    (tmp_path / "00003.png").write_bytes(raw[: len(raw) // 2])
    c = VideoProcessor(detector=SyntheticDetector(2, size=64), predictor=fake_predictor(), **kw)
    assert len(c.load_frames_from_folder(str(tmp_path))) in (6, 7)
    segs_c = c.run(frame_dir=str(tmp_path))
    assert sorted(segs_c) == list(range(6))


def test_remove_object_equals_never_having_added_it():
    """remove_object / clear_all_prompts_in_frame (sam2_video_predictor.py:1438-1549, :1061-1131): after the removal the
    state is the one of a run that never had the object - the stand-in stages treat batch rows independently, so every
    stored tensor and every later mask is equal bit for bit.  Reference arithmetic: golden e2e_remove (oracle + GPU tests)."""
    def start(objs):
        p = fake_predictor()
        st = p.init_state(_frames(6))
        for o in objs:
            p.add_new_points_or_box(st, 0, o, box=synthetic_box(o, 0, size=64))
        return p, st

    a, sa = start([0, 1, 2])
    a.add_new_points_or_box(sa, 3, 1, points=np.array([[30.0, 30.0]], np.float32), labels=np.array([1], np.int32))
    list(a.propagate_in_video(sa, start_frame_idx=0, max_frame_num_to_track=6))
    assert sorted(sa["output_dict"]["cond_frame_outputs"]) == [0, 3]
    # a correction click that is taken back: the frame shows the tracked masks again, nothing stays in the temp dict
    before = a._video_res(sa, sa["output_dict"]["non_cond_frame_outputs"][4]["pred_masks"])
    a.add_new_points_or_box(sa, 4, 0, points=np.array([[10.0, 12.0]], np.float32), labels=np.array([0], np.int32))
    t, ids, vr = a.clear_all_prompts_in_frame(sa, 4, 0)
    assert (t, list(ids)) == (4, [0, 1, 2]) and torch.equal(vr, before)
    assert not sa["temp_output_dict_per_obj"][0]["non_cond_frame_outputs"] and 4 not in sa["point_inputs_per_obj"][0]
    ids, updated = a.remove_object(sa, 1)
    assert list(ids) == [0, 2] and sorted(t for t, _ in updated) == [0, 3]
    assert all(m.shape == (2, 1, 64, 64) for _, m in updated)
    assert sa["obj_id_to_idx"] == {0: 0, 2: 1} and sa["obj_idx_to_id"] == {0: 0, 1: 2}
    od = sa["output_dict"]
    assert sorted(od["cond_frame_outputs"]) == [0] and sorted(od["non_cond_frame_outputs"]) == [1, 2, 3, 4, 5]   # 3 demoted
    assert 3 not in sa["frames_already_tracked"] and 3 not in sa["consolidated_frame_inds"]["cond_frame_outputs"]
    for name in ("point_inputs_per_obj", "mask_inputs_per_obj", "output_dict_per_obj", "temp_output_dict_per_obj"):
        assert sorted(sa[name]) == [0, 1]
    assert a.remove_object(sa, 77) == ([0, 2], [])
    with pytest.raises(RuntimeError, match="77"):
        a.remove_object(sa, 77, strict=True)

    b, sb = start([0, 2])
    list(b.propagate_in_video(sb, start_frame_idx=0, max_frame_num_to_track=6))
    for f in ("maskmem_features", "pred_masks", "obj_ptr", "object_score_logits"):       # the shared conditioning frame
        assert torch.equal(od["cond_frame_outputs"][0][f], sb["output_dict"]["cond_frame_outputs"][0][f]), f
    for key in ("cond_frame_outputs", "non_cond_frame_outputs"):                          # per-object views are re-cut
        for t, out in od[key].items():
            assert out["pred_masks"].shape[0] == 2 and out["maskmem_features"].shape[0] == 2
            for i in range(2):
                assert torch.equal(sa["output_dict_per_obj"][i][key][t]["obj_ptr"], out["obj_ptr"][i:i + 1])
    # propagation after the removal: every frame but 0 is tracked again with 2 objects and a bank that no longer holds
    # object 1's conditioning frame 3 - the run equals run B frame by frame
    n0 = _tracked(a)
    ya = {t: m.clone() for t, _, m in a.propagate_in_video(sa, start_frame_idx=0, max_frame_num_to_track=6)}
    yb = {t: m.clone() for t, _, m in b.propagate_in_video(sb, start_frame_idx=0, max_frame_num_to_track=6)}
    assert _tracked(a) == n0 + 5 and 3 in sa["frames_already_tracked"]
    for t in range(6):
        assert ya[t].shape == (2, 1, 64, 64) and torch.equal(ya[t], yb[t]), t

    # the last object: remove_object == reset_state; clearing the only prompt of the only conditioning frame resets tracking
    c, sc = start([5])
    assert c.remove_object(sc, 5) == ([], [])
    assert not sc["obj_id_to_idx"] and not sc["output_dict"]["cond_frame_outputs"]
    d, sd = start([5])
    list(d.propagate_in_video(sd, start_frame_idx=0, max_frame_num_to_track=3))
    assert d.clear_all_prompts_in_frame(sd, 0, 5, need_output=False) is None
    assert not sd["tracking_has_started"] and not sd["output_dict"]["non_cond_frame_outputs"] and sd["obj_ids"] == [5]
