"""The C-ABI library builds for gfx950 and exports every symbol include/detsam2_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    hdr = open(os.path.join(ROOT, "include", "detsam2_hip.h")).read()
    declared = set(re.findall(r"\b(ds2_[a-z0-9_]+)\s*\(", hdr))
    from det_sam2_amd import _capi
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ds2_abi_version() == 1


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from det_sam2_amd.hip_model import HipOps
    with pytest.raises(RuntimeError):
        HipOps("cuda:0")
