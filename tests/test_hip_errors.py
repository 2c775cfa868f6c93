"""Error behaviour at the boundary (SURVEY 8b "error conventions"): bad arguments come back as error codes with a
message (Python: Ds2Error, a RuntimeError like the reference's native op), never as crashes or silent garbage; the
predictor raises the reference's exception types."""
import numpy as np
import pytest
import torch

from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
TINY = "sam2.1_hiera_t"


@pytest.fixture(scope="module")
def hm():
    from det_sam2_amd.hip_model import HipSam2
    cfg = resolve_config(TINY)
    return HipSam2(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=2)


def test_c_abi_rejects_bad_arguments(hm):
    from det_sam2_amd._capi import Ds2Error
    d = hm.device
    curr = torch.zeros(4096, 256, device=d)
    mem = torch.zeros(1, 4096 + 10, 64, device=d)
    # the stages are PyTorch custom ops over the C-ABI: an error code surfaces as c10::Error = RuntimeError with the
    # library's message (as the reference's native op does, connected_components.cu:215-227)
    with pytest.raises(RuntimeError, match="multiple of 4096"):         # Nk - num_obj_ptr_tokens must be whole frames
        hm.memory_attention(1, curr, mem, mem, 4)
    f0, f1, f2 = torch.zeros(65536, 32, device=d), torch.zeros(16384, 64, device=d), torch.zeros(1, 4096, 256, device=d)
    with pytest.raises(RuntimeError, match="bad prompt"):               # more points than the entry point accepts (256)
        hm.sam_heads(1, f2, f0, f1, torch.zeros(1, 300, 2, device=d), torch.zeros(1, 300, dtype=torch.int32, device=d), False)
    q = torch.zeros(1, 8, 40, device=d)
    with pytest.raises(Ds2Error, match="unsupported head dims"):
        hm.op_attention(q, q, q, 1, 1.0)
    with pytest.raises(ValueError):
        hm.set_precision("fp8")
    with pytest.raises(Ds2Error):
        from det_sam2_amd import _capi
        _capi.check(hm.lib.ds2_set_precision(7), "ds2_set_precision")
    # the handle still works after errors
    out = hm.memory_attention(1, curr, torch.zeros(1, 4096, 64, device=d), torch.zeros(1, 4096, 64, device=d), 0)
    assert torch.isfinite(out).all()


def test_predictor_raises_like_the_reference():
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    cfg = resolve_config(TINY)
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=2)
    st = pred.init_state([synthetic_frame(0), synthetic_frame(1)])
    with pytest.raises(RuntimeError, match="[Nn]o points"):            # sam2_video_predictor.py:943-944
        next(pred.propagate_in_video(st))
    with pytest.raises(ValueError):                                   # :363-366
        pred.add_new_points_or_box(st, 0, 1)
    with pytest.raises(ValueError):
        pred.add_new_points_or_box(st, 0, 1, points=np.zeros((1, 2), np.float32))
    with pytest.raises(ValueError):                                   # :384-388 box needs clear_old_points
        pred.add_new_points_or_box(st, 0, 1, box=np.array([1, 2, 30, 40], np.float32), clear_old_points=False)
    with pytest.raises(NotImplementedError):                          # frame sources outside the hot path (misc.py:292-303)
        pred.init_state("/some/video.mp4")
    # the state is still usable
    pred.add_new_points_or_box(st, 0, 1, box=np.array([100, 100, 400, 400], np.float32))
    outs = list(pred.propagate_in_video(st))
    assert [o[0] for o in outs] == [0, 1]
    # prompts on tracked frames are corrections (tests/test_hip_correct.py): stored as non-conditioning temp outputs
    pred.add_new_mask(st, 0, 1, np.zeros((1024, 1024), bool))
    pred.add_new_points_or_box(st, 1, 1, box=np.array([100, 100, 400, 400], np.float32))
    tmp = st["temp_output_dict_per_obj"][0]["non_cond_frame_outputs"]
    assert sorted(tmp) == [0, 1]
    # wrong tensor types never reach the kernels
    with pytest.raises(RuntimeError, match="must be"):
        pred.hip.ops.memory_encoder(pred.hip._h, 1, torch.zeros(4096, 256, device="cuda:0", dtype=torch.float16),
                                    torch.zeros(1, 256, 256, device="cuda:0"), torch.zeros(1, device="cuda:0"), False)
    with pytest.raises(NotImplementedError):                          # a CPU tensor finds no kernel: there is no CPU path
        pred.hip.ops.fill_holes(torch.zeros(1, 1, 8, 8), 8)
