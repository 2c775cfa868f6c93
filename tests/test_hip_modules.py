"""Module-level boundary (det_sam2_amd.modules: nn.Modules with the reference's signatures and layouts) against the
REFERENCE'S OWN outputs: the seeded inputs of oracle/make_goldens.py:l1_inputs go in, tests/golden/l1_<config>.npz (written
by the reference modules on those inputs) is the expectation - no oracle in between.  All four SAM 2.1 configs."""
import os

import numpy as np
import pytest
import torch

from _util import record
from det_sam2_amd.config import resolve_config
from det_sam2_amd.weights import synthetic_state_dict
from oracle.make_goldens import l1_inputs

pytestmark = pytest.mark.gpu
CONFIGS = ["sam2.1_hiera_t", "sam2.1_hiera_s", "sam2.1_hiera_b+", "sam2.1_hiera_l"]
TOL = {"fp32": 2e-5, "bf16x3": 1e-3, "bf16x3k": 1e-3}


def _rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


@pytest.fixture(scope="module", params=CONFIGS)
def mods(request):
    from det_sam2_amd.modules import build_modules
    cfg = resolve_config(request.param)
    return request.param, build_modules(cfg, synthetic_state_dict(cfg, 0), "cuda:0", max_batch=2)


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16x3k"])
def test_modules_match_reference_fixtures(mods, prec, golden_dir):
    name, m = mods
    if prec != "bf16x3k" and name not in ("sam2.1_hiera_t", "sam2.1_hiera_l"):
        pytest.skip("the non-default modes are covered on the smallest and the largest config")
    m["hip"].set_precision(prec)
    g = np.load(os.path.join(golden_dir, f"l1_{name}.npz"))
    x = l1_inputs()
    tol = TOL[prec]
    errs = {}
    # ---- SAM2Base.forward_image
    out = m["image_encoder"](x["img"])
    assert set(out) == {"vision_features", "vision_pos_enc", "backbone_fpn"}
    assert [tuple(f.shape) for f in out["backbone_fpn"]] == [(1, 32, 256, 256), (1, 64, 128, 128), (1, 256, 64, 64)]
    for i, f in enumerate(out["backbone_fpn"]):
        errs[f"fpn{i}"] = _rel(f[0, ::4, ::8, ::8].cpu(), g[f"fpn{i}"])
    errs["pos2"] = _rel(out["vision_pos_enc"][2][0, ::8, ::8, ::8].cpu(), g["pos2"])
    # ---- MemoryAttention.forward  (sequence-first tensors, per-object tokens, explicit positions)
    ma = m["memory_attention"](curr=[x["curr"]], curr_pos=[x["curr_pos"]], memory=x["mem"], memory_pos=x["mem_pos"],
                               num_obj_ptr_tokens=8)
    assert tuple(ma.shape) == (4096, 2, 256)
    errs["memattn"] = _rel(ma[::32, :, ::4].cpu(), g["memattn"])
    # ---- MemoryEncoder.forward
    me = m["memory_encoder"](x["pix"], x["masks"], skip_mask_sigmoid=True)
    assert tuple(me["vision_features"].shape) == (2, 64, 64, 64)
    errs["memenc"] = _rel(me["vision_features"][:, ::2, ::4, ::4].cpu(), g["memenc"])
    errs["memenc_pos"] = _rel(me["vision_pos_enc"][0][0, :, ::8, ::8].cpu(), g["memenc_pos"])
    # ---- SAM2Base._forward_sam_heads (prompt encoder + mask decoder + selection + pointer)
    fs = m["sam_heads"](backbone_features=x["emb"], point_inputs=None, mask_inputs=None, high_res_features=[x["hr0"], x["hr1"]],
                        multimask_output=True)
    errs["heads_low"] = _rel(fs[3][:, :, ::4, ::4].cpu(), g["heads_low"])
    errs["heads_high"] = _rel(fs[4][:, :, ::16, ::16].cpu(), g["heads_high"])
    errs["heads_ptr"] = _rel(fs[5].cpu(), g["heads_ptr"])
    errs["heads_obj"] = _rel(fs[6].cpu(), g["heads_obj"])
    # ---- PromptEncoder.forward / get_dense_pe / mask_input_size (prompt_encoder.py:134-171,64-71,49)
    pe_mod, md_mod = m["sam_prompt_encoder"], m["sam_mask_decoder"]
    assert tuple(pe_mod.mask_input_size) == (256, 256)
    sp, de = pe_mod(points=(x["coords"], x["labels"]), boxes=None, masks=x["mask_prompt"])
    assert tuple(sp.shape) == (2, 3, 256) and tuple(de.shape) == (2, 256, 64, 64)
    errs["sparse"] = _rel(sp.cpu(), g["sparse"])
    errs["dense"] = _rel(de[:, ::8, ::4, ::4].cpu(), g["dense"])
    dpe = pe_mod.get_dense_pe()
    assert tuple(dpe.shape) == (1, 256, 64, 64)
    errs["dense_pe"] = _rel(dpe[0, ::8, ::4, ::4].cpu(), g["dense_pe"])
    # boxes: the two corners as points labelled 2 / 3 without the padding point (prompt_encoder.py:106-116,158) - the same
    # embeddings the (coords, labels = [2, 3]) point form gives before its padding token
    spb, _ = pe_mod(points=None, boxes=x["coords"].reshape(2, 4), masks=None)
    assert tuple(spb.shape) == (2, 2, 256) and torch.equal(spb, sp[:, :2])
    # no points and no boxes: an EMPTY sparse tensor [B,0,256] + the dense embedding, as the reference returns
    # (prompt_encoder.py:155-171) - with a mask prompt the mask_downscaling output, without one the no_mask_embed broadcast
    sp0, de0 = pe_mod(points=None, boxes=None, masks=x["mask_prompt"])
    assert tuple(sp0.shape) == (2, 0, 256) and torch.equal(de0, de)
    sp1, de1 = pe_mod(points=None, boxes=None, masks=None)
    assert tuple(sp1.shape) == (1, 0, 256) and tuple(de1.shape) == (1, 256, 64, 64)
    assert torch.equal(de1[0], pe_mod(points=(x["coords"], x["labels"]), boxes=None, masks=None)[1][0])
    # ---- MaskDecoder.forward (mask_decoder.py:105-161) on the prompt encoder's own outputs, both multimask settings
    s_, d_ = pe_mod(points=(x["coords"], x["labels"]), boxes=None, masks=None)
    for mm in (True, False):
        r = md_mod(image_embeddings=x["emb"], image_pe=dpe, sparse_prompt_embeddings=s_, dense_prompt_embeddings=d_,
                   multimask_output=mm, repeat_image=False, high_res_features=[x["hr0"], x["hr1"]])
        n = 3 if mm else 1
        assert tuple(r[0].shape) == (2, n, 256, 256) and tuple(r[1].shape) == (2, n) and tuple(r[2].shape) == (2, n, 256) and tuple(r[3].shape) == (2, 1)
        errs[f"dec{int(mm)}_masks"] = _rel(r[0][:, :, ::4, ::4].cpu(), g[f"dec{int(mm)}_masks"])
        errs[f"dec{int(mm)}_iou"] = _rel(r[1].cpu(), g[f"dec{int(mm)}_iou"])
        errs[f"dec{int(mm)}_tok"] = _rel(r[2].cpu(), g[f"dec{int(mm)}_tok"])
        errs[f"dec{int(mm)}_obj"] = _rel(r[3].cpu(), g[f"dec{int(mm)}_obj"])
    record("modules_vs_reference", config=name, prec=prec, **errs)
    for k, e in errs.items():
        bound = (1e-6 if k in ("pos2", "memenc_pos", "dense_pe", "sparse") else
                 (tol * (3 if k.startswith("fpn") or k.startswith("heads") or k.startswith("dec") else 1)))
        assert e <= bound, (name, prec, k, e, errs)
