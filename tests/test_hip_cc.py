"""A14 on the GPU: HIP connected components / hole filling (through the C-ABI) vs oracle/cc.py - bit-exact (integer
work), including odd sizes, empty / full masks, diagonals, long snakes (deep union-find chains) and the predictor
path with fill_hole_area=8."""
import numpy as np
import pytest
import torch

from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import SyntheticDetector, synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict
from oracle.cc import connected_components, fill_holes_in_mask_scores

from _util import record

pytestmark = pytest.mark.gpu


def _snake(H, W):
    m = np.zeros((H, W), np.uint8)
    for y in range(0, H, 2):
        m[y, :] = 1
        if y + 1 < H:
            m[y + 1, (W - 1) if (y // 2) % 2 == 0 else 0] = 1
    return m


def _cases():
    rng = np.random.default_rng(1)
    out = {"empty": np.zeros((2, 1, 8, 8), np.uint8), "full": np.ones((1, 1, 33, 17), np.uint8),
           "one_px": np.ones((1, 1, 1, 1), np.uint8), "diag": np.eye(64, dtype=np.uint8)[None, None],
           "checker": (np.indices((40, 40)).sum(0) % 2).astype(np.uint8)[None, None],
           "snake": _snake(255, 253)[None, None], "snake_even": _snake(256, 256)[None, None]}
    for p in (0.05, 0.4, 0.5928, 0.9):
        out[f"rand{p}"] = (rng.random((3, 1, 37, 53)) < p).astype(np.uint8)
    out["rand_batch16_256"] = (rng.random((16, 1, 256, 256)) < 0.45).astype(np.uint8)
    return out


@pytest.mark.parametrize("name", sorted(_cases()))
def test_connected_components_match_oracle(name):
    from det_sam2_amd.misc import get_connected_components
    m = _cases()[name]
    lab, cnt = get_connected_components(torch.from_numpy(m).cuda())
    torch.cuda.synchronize()
    l0, c0 = connected_components(m)
    assert lab.dtype == torch.int32 and cnt.dtype == torch.int32 and tuple(lab.shape) == m.shape
    assert np.array_equal(cnt.cpu().numpy(), c0)
    assert np.array_equal(lab.cpu().numpy(), l0)      # both number a component by its smallest raster index + 1


def test_connected_components_bool_input_and_errors():
    from det_sam2_amd.misc import get_connected_components
    m = torch.rand(2, 1, 16, 16, device="cuda") > 0.5
    lab, cnt = get_connected_components(m)
    l0, c0 = connected_components(m.cpu().numpy())
    assert np.array_equal(cnt.cpu().numpy(), c0)
    with pytest.raises(ValueError):
        get_connected_components(torch.zeros(4, 4, device="cuda"))
    with pytest.raises(RuntimeError):
        get_connected_components(torch.zeros(1, 1, 4, 4))


@pytest.mark.parametrize("max_area", [1, 8, 100])
def test_fill_holes_matches_oracle(max_area):
    from det_sam2_amd.misc import fill_holes_in_mask_scores as hip_fill
    rng = np.random.default_rng(max_area)
    yy, xx = np.mgrid[0:256, 0:256]
    s = np.stack([8.0 - np.hypot(yy - 128 - 9 * i, xx - 100 - 5 * i) / (6 + i) for i in range(16)]).astype(np.float32)
    s += rng.normal(0, 2.5, s.shape).astype(np.float32)          # salt: many small holes and islands
    s[rng.random(s.shape) < 0.01] = 0.0                            # exact zeros are background
    s = s[:, None]
    got = hip_fill(torch.from_numpy(s).cuda(), max_area).cpu().numpy()
    ref = fill_holes_in_mask_scores(s, max_area)
    n_filled = int((ref != s).sum())
    record("fill_holes", max_area=max_area, filled=n_filled)
    assert n_filled > 0
    assert np.array_equal(got, ref)
    with pytest.raises(AssertionError):
        hip_fill(torch.from_numpy(s).cuda(), 0)


def test_predictor_with_hole_filling_matches_oracle(golden_dir):
    """tiny model, 2 objects, 4 frames, fill_hole_area=8 on both sides (the reference's GPU behaviour).  The oracle's masks are the
    committed fixture oracle_fill8_tiny.npz (oracle/make_oracle_fixtures.py fill8_tiny; DS2_SLOW_ORACLE=1 runs the oracle alongside)."""
    import os
    from det_sam2_amd.det_sam2_RT import VideoProcessor
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config("sam2.1_hiera_t")
    sd = synthetic_state_dict(cfg, 0)
    kw = dict(frame_buffer_size=4, detect_interval=4, max_frame_num_to_track=4, max_inference_state_frames=-1)
    pred = SAM2VideoPredictor(cfg, sd, "cuda:0", max_batch=4, fill_hole_area=8)
    vp = VideoProcessor(model_cfg="sam2.1_hiera_t", detector=SyntheticDetector(2), skip_classes=set(), predictor=pred, **kw)
    for t in range(4):
        vp.process_frame(t, synthetic_frame(t))
    if os.environ.get("DS2_SLOW_ORACLE"):
        from oracle.video_processor import OracleVideoProcessor
        ovp = OracleVideoProcessor(sd, cfg, SyntheticDetector(2), skip_classes=set(), fill_hole_area=8, **kw)
        ovp0 = OracleVideoProcessor(sd, cfg, SyntheticDetector(2), skip_classes=set(), fill_hole_area=0, **kw)
        for t in range(4):
            f = synthetic_frame(t)
            ovp.process_frame(t, f)
            ovp0.process_frame(t, f)
        want = {t: {o: np.asarray(ovp.video_segments[t][o]).astype(bool) for o in ovp.video_segments[t]} for t in range(4)}
        changed = sum(int((want[t][o] != np.asarray(ovp0.video_segments[t][o]).astype(bool)).sum()) for t in range(4) for o in want[t])
    else:
        g = np.load(os.path.join(golden_dir, "oracle_fill8_tiny.npz"))
        want, changed = {}, int(g["pixels_changed_by_filling"])
        for t in range(4):
            objs = [int(o) for o in g[f"objs{t}"]]
            bits = np.unpackbits(g[f"bits{t}"])[: len(objs) * 1024 * 1024].reshape(len(objs), 1, 1024, 1024).astype(bool)
            want[t] = {o: bits[j] for j, o in enumerate(objs)}
    worst = 0.0
    for t in range(4):
        assert sorted(vp.video_segments[t]) == sorted(want[t])
        for o in vp.video_segments[t]:
            a, b = np.asarray(vp.video_segments[t][o]).astype(bool), want[t][o]
            u = (a | b).sum()
            worst = max(worst, 1.0 - ((a & b).sum() / u if u else 1.0))
    record("e2e_fill8", one_minus_iou=worst, pixels_changed_by_filling=changed)
    assert changed > 0            # the test exercises the step
    assert worst <= 1e-3, worst
