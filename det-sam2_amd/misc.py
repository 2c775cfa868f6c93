"""Mask post-processing ops of the hot path, mirroring ``sam2/utils/misc.py`` of the reference.

* ``get_connected_components`` (misc.py:48-61 -> sam2/_C ``get_connected_componnets``,
  csrc/connected_components.cu:213-289)
* ``fill_holes_in_mask_scores`` (misc.py:365-393)

Both run on the GPU as PyTorch custom ops over the C-ABI (``ds2_connected_components`` / ``ds2_fill_holes``).  Unlike the reference,
which swallows every exception of its CUDA extension and silently skips hole filling (misc.py:389-391), a missing
library or a non-GPU tensor is an error here.
"""
from __future__ import annotations

import torch

from . import _capi


def get_connected_components(mask: torch.Tensor):
    """mask [N,1,H,W] (non-zero = foreground) on the GPU -> (labels int32 [N,1,H,W], counts int32 [N,1,H,W]).
    Calls ``torch.ops.det_sam2.get_connected_componnets`` - the reference's op name, spelling included
    (connected_components.cu:284-289)."""
    if mask.dim() != 4 or mask.shape[1] != 1:
        raise ValueError(f"mask must be [N,1,H,W], got {tuple(mask.shape)}")
    if not mask.is_cuda:
        raise RuntimeError("get_connected_components: GPU tensor required (there is no CPU path)")
    ops = _capi.load_torch_ops()
    m = mask.to(torch.uint8).contiguous()
    if m.numel() == 0:
        z = torch.empty(m.shape, dtype=torch.int32, device=m.device)
        return z, z.clone()
    labels, counts = ops.get_connected_componnets(m)
    return labels, counts


def fill_holes_in_mask_scores(mask: torch.Tensor, max_area: int):
    """Background components (score <= 0) of area <= max_area become score 0.1.  Returns a new tensor."""
    assert max_area > 0, "max_area must be positive"
    if not mask.is_cuda:
        raise RuntimeError("fill_holes_in_mask_scores: GPU tensor required (there is no CPU path)")
    ops = _capi.load_torch_ops()
    x = mask.to(torch.float32).contiguous()
    if x.numel() == 0:
        return x.clone()
    return ops.fill_holes(x, int(max_area))
