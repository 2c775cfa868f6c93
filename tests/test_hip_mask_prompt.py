"""F3 on the GPU: add_new_mask / _use_mask_as_output (sam2_video_predictor.py:527-616, sam2_base.py:399-448) against a
golden produced by the reference predictor (oracle/make_goldens.py e2e_mask), plus the antialiased resize kernel against
torch's own F.interpolate (the arithmetic the reference calls)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import record
from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
TINY = "sam2.1_hiera_t"


@pytest.mark.parametrize("hin,win,hout,wout", [(1024, 1024, 256, 256), (384, 512, 1024, 1024), (1080, 1920, 1024, 1024),
                                                (540, 960, 256, 256), (7, 5, 16, 3), (256, 256, 256, 256)])
def test_resize_aa_matches_torch(hin, win, hout, wout):
    from det_sam2_amd.hip_model import HipOps
    ops = HipOps("cuda:0")
    g = torch.Generator().manual_seed(hin * 7 + wout)
    x = torch.rand(2, hin, win, generator=g)
    ref = F.interpolate(x[:, None], size=(hout, wout), mode="bilinear", align_corners=False, antialias=True)[:, 0]
    from det_sam2_amd.hip_model import HipSam2
    got = HipSam2.resize_aa(ops, x.cuda(), hout, wout).cpu()
    err = float((got - ref).abs().max())
    # binary masks, affine source map and the >= 0.5 threshold of add_new_mask
    m = (x > 0.6).float()
    refb = (F.interpolate(m[:, None], size=(hout, wout), mode="bilinear", align_corners=False, antialias=True)[:, 0] >= 0.5).float()
    gotb = HipSam2.resize_aa(ops, m.cuda(), hout, wout, threshold=0.5).cpu()
    flips = int((gotb != refb).sum())
    refl = F.interpolate(m[:, None] * 20.0 - 10.0, size=(hout, wout), mode="bilinear", align_corners=False, antialias=True)[:, 0]
    gotl = HipSam2.resize_aa(ops, m.cuda(), hout, wout, 20.0, -10.0).cpu()
    errl = float((gotl - refl).abs().max())
    record("resize_aa", shape=f"{hin}x{win}->{hout}x{wout}", err=err, flips=flips, err_logits=errl)
    assert err <= 2e-6 and errl <= 2e-5 and flips <= max(1, refb.numel() // 100000), (err, errl, flips)
    if (hin, win) == (1024, 1024):
        assert errl == 0.0          # exact 4x downscale of a 0/1 mask: dyadic weights, every product and sum exact


@pytest.mark.parametrize("prec", ["fp32", "bf16x3k"])
def test_add_new_mask_matches_reference(golden_dir, prec):
    from det_sam2_amd.sam2_video_predictor import SAM2VideoPredictor
    from oracle.make_goldens import mask_prompts
    g = np.load(os.path.join(golden_dir, "e2e_mask.npz"))
    cfg = resolve_config(TINY)
    pred = SAM2VideoPredictor(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=4)
    pred.hip.set_precision(prec)
    m0, m1 = mask_prompts()
    st = pred.init_state([synthetic_frame(t) for t in range(4)])
    for oid, m in ((0, m0), (1, m1), (2, np.zeros((1024, 1024), bool))):
        _, ids, vr = pred.add_new_mask(st, 0, oid, m)
    assert list(ids) == [0, 1, 2]
    assert np.array_equal(np.packbits((vr > 0).cpu().numpy()), g["prompt_bits2"])
    worst, worst_logit, i = 0.0, 0.0, 0
    for t, ids, logits in pred.propagate_in_video(st, start_frame_idx=0, max_frame_num_to_track=4, reverse=False):
        od = st["output_dict"]
        key = "cond_frame_outputs" if t in od["cond_frame_outputs"] else "non_cond_frame_outputs"
        assert t == int(g["frames"][i])
        low = od[key][t]["pred_masks"].cpu().numpy()
        d = float(np.abs(low - g["low"][i]).max())
        if i == 0:
            assert d == 0.0                                                       # the mask IS the output on the prompted frame
            assert np.array_equal(od[key][t]["object_score_logits"].cpu().numpy(), g["obj_score0"])
            assert np.abs(od[key][t]["obj_ptr"].cpu().numpy() - g["obj_ptr0"]).max() <= (1e-4 if prec == "fp32" else 2e-3)
        worst_logit = max(worst_logit, d)
        ref = np.unpackbits(g["bits"][i]).reshape(3, 1, 1024, 1024).astype(bool)
        bits = (logits > 0).cpu().numpy()
        for o in range(3):
            inter, union = (bits[o] & ref[o]).sum(), (bits[o] | ref[o]).sum()
            worst = max(worst, 1.0 - (inter / union if union else 1.0))
        i += 1
    assert i == 4
    record("e2e_mask", prec=prec, one_minus_iou=worst, max_abs_dlogit=worst_logit)
    assert worst <= 1e-3 and worst_logit <= (5e-3 if prec == "fp32" else 5e-2), (worst, worst_logit)
