"""Host-side constant tables (det_sam2_amd.constants, numpy) vs the oracle's torch formulation. CPU only."""
import numpy as np
import torch

from det_sam2_amd import constants as K
from det_sam2_amd.config import resolve_config
from det_sam2_amd.synth import synthetic_frame
from det_sam2_amd.weights import synthetic_state_dict
from oracle import modeling as M
from oracle.predictor import load_frames


def test_constants_match_oracle():
    for name in ("sam2.1_hiera_t", "sam2.1_hiera_b+"):
        cfg = resolve_config(name)
        sd = synthetic_state_dict(cfg, 0)
        c = K.model_constants(cfg, {k: v.numpy() for k, v in sd.items()})
        pe = M.hiera_pos_embed(sd, "image_encoder.trunk", (256, 256))[0].reshape(65536, -1).numpy()
        assert np.abs(c["#pos_embed"] - pe).max() < 2e-6
        cis = M.axial_cis(256, 64, 64)
        assert np.abs(c["#rope_cis"][..., 0] - cis.real.numpy()).max() < 1e-5
        assert np.abs(c["#rope_cis"][..., 1] - cis.imag.numpy()).max() < 1e-5
        vp = M.sine_pos_2d(256, 64, 64).permute(1, 2, 0).reshape(4096, 256).numpy()
        assert np.abs(c["#vision_pos"] - vp).max() < 1e-5
        mp = M.sine_pos_2d(64, 64, 64).permute(1, 2, 0).reshape(4096, 64).numpy()
        assert np.abs(c["#maskmem_pos"] - mp).max() < 1e-5
        dp = M.dense_pe(sd, cfg)[0].permute(1, 2, 0).reshape(4096, 256).numpy()
        assert np.abs(c["#dense_pe"] - dp).max() < 2e-5
        e = M.sine_pe_1d(torch.tensor([3.0 / 15]), 256).numpy()[0]
        v = (3.0 / 15) / c["#ptr_dim_t"]
        assert np.abs(np.concatenate([np.sin(v), np.cos(v)]) - e).max() < 1e-6


def test_ingest_lut_is_bit_exact_with_reference_storage_chain():
    lut = K.ingest_lut()
    # every byte value in every channel
    fr = np.zeros((1024, 1024, 3), np.uint8)
    fr[:256, 0, :] = np.arange(256, dtype=np.uint8)[:, None]
    imgs, _, _ = load_frames([fr])
    got = imgs[0, :, :256, 0].numpy().view(np.uint16)
    assert np.array_equal(got, lut)
    imgs, _, _ = load_frames([synthetic_frame(3)])
    f = synthetic_frame(3)
    mine = np.stack([lut[c][f[..., c]] for c in range(3)])
    assert np.array_equal(imgs[0].numpy().view(np.uint16), mine)
