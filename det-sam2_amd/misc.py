"""Mask post-processing ops of the hot path, mirroring ``sam2/utils/misc.py`` of the reference.

* ``get_connected_components`` (misc.py:48-61 -> sam2/_C ``get_connected_componnets``,
  csrc/connected_components.cu:213-289)
* ``fill_holes_in_mask_scores`` (misc.py:365-393)

Both run on the GPU through the C-ABI (``ds2_connected_components`` / ``ds2_fill_holes``).  Unlike the reference,
which swallows every exception of its CUDA extension and silently skips hole filling (misc.py:389-391), a missing
library or a non-GPU tensor is an error here.
"""
from __future__ import annotations

import torch

from . import _capi


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def get_connected_components(mask: torch.Tensor):
    """mask [N,1,H,W] (non-zero = foreground) on the GPU -> (labels int32 [N,1,H,W], counts int32 [N,1,H,W])."""
    if mask.dim() != 4 or mask.shape[1] != 1:
        raise ValueError(f"mask must be [N,1,H,W], got {tuple(mask.shape)}")
    if not mask.is_cuda:
        raise RuntimeError("get_connected_components: GPU tensor required (there is no CPU path)")
    lib = _capi.load()
    m = mask.to(torch.uint8).contiguous()
    N, _, H, W = m.shape
    labels = torch.empty((N, 1, H, W), dtype=torch.int32, device=m.device)
    counts = torch.empty_like(labels)
    if N == 0 or H == 0 or W == 0:
        return labels, counts
    work = torch.empty((2 * N * H * W,), dtype=torch.int32, device=m.device)
    _capi.check(lib.ds2_connected_components(_p(m), N, H, W, _p(labels), _p(counts), _p(work), _stream()),
                "ds2_connected_components")
    return labels, counts


def fill_holes_in_mask_scores(mask: torch.Tensor, max_area: int):
    """Background components (score <= 0) of area <= max_area become score 0.1.  Returns a new tensor."""
    assert max_area > 0, "max_area must be positive"
    if not mask.is_cuda:
        raise RuntimeError("fill_holes_in_mask_scores: GPU tensor required (there is no CPU path)")
    lib = _capi.load()
    out = mask.to(torch.float32).contiguous().clone()
    if out.numel() == 0:
        return out
    H, W = out.shape[-2], out.shape[-1]
    N = out.numel() // (H * W)
    work = torch.empty((3 * N * H * W,), dtype=torch.int32, device=out.device)
    _capi.check(lib.ds2_fill_holes(_p(out), N, H, W, int(max_area), _p(work), _stream()), "ds2_fill_holes")
    return out
