"""F2: the versioned preload-bank file (det_sam2_amd.bank_io) - CPU-only checks of the format itself.  The GPU tests
(tests/test_hip_e2e.py) run the preload scenario of the reference golden through it."""
import pickle
from collections import OrderedDict

import numpy as np
import pytest
import torch

from det_sam2_amd import bank_io


def _entry(B, g, with_mem=True):
    return {"maskmem_features": torch.randn(B, 4096, 64, generator=g).to(torch.bfloat16) if with_mem else None,
            "maskmem_pos_enc": None, "pred_masks": torch.randn(B, 1, 256, 256, generator=g),
            "obj_ptr": torch.randn(B, 256, generator=g), "object_score_logits": torch.randn(B, 1, generator=g)}


def _state(B=3):
    g = torch.Generator().manual_seed(5)
    st = {"num_frames": 7, "video_height": 540, "video_width": 960, "obj_ids": [4, 9, 2][:B], "images_idx": list(range(7)),
          "cached_features": {}, "output_dict": {"cond_frame_outputs": {0: _entry(B, g), 5: _entry(B, g)},
                                                 "non_cond_frame_outputs": {1: _entry(B, g), 2: _entry(B, g, with_mem=False)}}}
    return st, g


def test_ds2bank_round_trip(tmp_path):
    st, g = _state()
    fpn2 = {0: torch.randn(4096, 256, generator=g), 5: torch.randn(4096, 256, generator=g)}
    path = str(tmp_path / "bank.ds2")
    hdr = bank_io.save_bank(path, st, "sam2.1_hiera_t", lambda t: fpn2.get(t))
    assert open(path, "rb").read(8) == bank_io.MAGIC
    assert [e["frame"] for e in hdr["entries"]] == [0, 5, 1]          # the entry without memory features is not stored
    got = bank_io.load_bank(path)
    assert got["num_frames"] == 7 and (got["video_height"], got["video_width"]) == (540, 960)
    assert got["obj_ids"] == [4, 9, 2] and list(got["obj_id_to_idx"].items()) == [(4, 0), (9, 1), (2, 2)]
    assert len(got["images"]) == 0 and got["images_idx"] == []        # no frames travel
    assert got["consolidated_frame_inds"]["cond_frame_outputs"] == {0, 5}
    for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
        for t, out in got["output_dict"][key].items():
            for k in bank_io.ENTRY_TENSORS:
                ref = st["output_dict"][key][t][k]
                assert out[k].dtype == ref.dtype and torch.equal(out[k], ref), (key, t, k)
    for t in (0, 5):
        assert torch.equal(got["preload_fpn2"][t], fpn2[t])


def test_ds2bank_rejects_corruption(tmp_path):
    st, _ = _state(1)
    path = str(tmp_path / "bank.ds2")
    bank_io.save_bank(path, st, "sam2.1_hiera_t")
    raw = bytearray(open(path, "rb").read())
    with open(path, "wb") as f:
        f.write(raw[: len(raw) // 2])                                  # truncated payload
    with pytest.raises(ValueError):
        bank_io.load_bank(path)


def test_reference_layout_pickle_is_converted(tmp_path):
    """A bank pickled by the reference holds maskmem_features [B,64,64,64] (channel-major) + maskmem_pos_enc lists +
    torch.device objects (det_sam2_RT.py:489-497)."""
    g = torch.Generator().manual_seed(1)
    f = torch.randn(2, 64, 64, 64, generator=g).to(torch.bfloat16)
    st = {"images": torch.zeros(3, 3, 8, 8, dtype=torch.float16), "num_frames": 3, "images_idx": [0, 1, 2],
          "video_height": 1024, "video_width": 1024, "device": torch.device("cpu"), "storage_device": torch.device("cpu"),
          "obj_id_to_idx": OrderedDict([(7, 0), (8, 1)]), "obj_idx_to_id": OrderedDict([(0, 7), (1, 8)]), "obj_ids": [7, 8],
          "output_dict": {"cond_frame_outputs": {0: {"maskmem_features": f, "maskmem_pos_enc": [torch.zeros(2, 64, 64, 64)],
                                                     "pred_masks": torch.randn(2, 1, 256, 256, generator=g),
                                                     "obj_ptr": torch.randn(2, 256, generator=g),
                                                     "object_score_logits": torch.randn(2, 1, generator=g)}},
                          "non_cond_frame_outputs": {}},
          "output_dict_per_obj": {}, "cached_features": {0: "junk"}, "consolidated_frame_inds": {"cond_frame_outputs": {0}, "non_cond_frame_outputs": set()}}
    path = str(tmp_path / "ref_bank.pkl")
    with open(path, "wb") as fh:
        pickle.dump(st, fh)
    got = bank_io.load_bank(path)
    e = got["output_dict"]["cond_frame_outputs"][0]
    assert e["maskmem_features"].shape == (2, 4096, 64) and e["maskmem_features"].dtype == torch.bfloat16
    assert torch.equal(e["maskmem_features"][1, 64 * 5 + 9], f[1, :, 5, 9])          # token (y=5,x=9) holds the channel vector
    assert e["maskmem_pos_enc"] is None and got["cached_features"] == {} and len(got["images"]) == 3


class _Evil:
    def __reduce__(self):
        import os
        return (os.system, ("echo pwned",))


def test_pickle_loader_is_restricted(tmp_path):
    path = str(tmp_path / "evil.pkl")
    with open(path, "wb") as fh:
        pickle.dump({"output_dict": _Evil()}, fh)
    with pytest.raises(pickle.UnpicklingError):
        bank_io.load_bank(path)


_FIRED = []


class _Marker:
    def __reduce__(self):
        return (_FIRED.append, ("nested payload executed",))


class _Nested:
    """Outer pickle that calls torch.storage._load_from_bytes on an inner, unrestricted pickle (in torch
    2.10 that function is torch.load(weights_only=False))."""
    def __reduce__(self):
        import torch.storage
        return (torch.storage._load_from_bytes, (pickle.dumps(_Marker()),))


def test_nested_storage_pickle_cannot_execute(tmp_path):
    path = str(tmp_path / "nested.pkl")
    with open(path, "wb") as fh:
        pickle.dump({"output_dict": _Nested()}, fh)
    _FIRED.clear()
    with pytest.raises(Exception):
        bank_io.load_bank(path)
    assert _FIRED == [], "the nested pickle inside _load_from_bytes ran an arbitrary callable"


def test_pickle_fallback_can_be_disabled(tmp_path, monkeypatch):
    st, _ = _state(1)
    path = str(tmp_path / "ref.pkl")
    with open(path, "wb") as fh:
        pickle.dump(st, fh)
    with pytest.raises(ValueError, match="not a DS2BANK"):
        bank_io.load_bank(path, allow_pickle=False)
    monkeypatch.setenv("DS2_BANK_ALLOW_PICKLE", "0")
    with pytest.raises(ValueError, match="not a DS2BANK"):
        bank_io.load_bank(path)


def test_ds2bank_marks_bank_frames_as_tracked(tmp_path):
    """Prompts on preload frames are corrections, not initial conditioning frames: a DS2BANK state lists its entries in
    frames_already_tracked like the reference's pickled state does."""
    st, _ = _state()
    path = str(tmp_path / "bank.ds2")
    bank_io.save_bank(path, st, "sam2.1_hiera_t")
    got = bank_io.load_bank(path)
    assert sorted(got["frames_already_tracked"]) == [0, 1, 5]
