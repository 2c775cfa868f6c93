"""Multi-GPU: pass-sharded data parallelism, one process per GPU.

A Det-SAM2 pass (newest ``frame_buffer_size`` frames + the previous window, tracked in reverse)
depends only on its <= 2*buffer frame images and on the *conditioning-frame* bank entries
(prompted frames); non-conditioning memories are recomputed inside the pass.  So passes shard
over ranks (pass k -> rank k mod N) with ONE exchange: the new cond-frame entry of every pass is
all-gathered (RCCL over xGMI on GPUs; ~12 MiB at 16 objects), because pass k+1 needs the cond entry
that rank k produced for the frame the two windows share.  No other data-path collective exists.

Only ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.distributed as dist

ENTRY_FIELDS = ("maskmem_features", "pred_masks", "obj_ptr", "object_score_logits")


def pass_owner(pass_idx: int, world_size: int) -> int:
    return pass_idx % world_size


def passes_of_rank(num_passes: int, world_size: int, rank: int) -> List[int]:
    return [k for k in range(num_passes) if pass_owner(k, world_size) == rank]


def pass_window(pass_idx: int, frame_buffer_size: int, max_frame_num_to_track: int):
    """Frames touched by pass k: it starts at the newest frame and tracks in reverse
    (det_sam2_RT.py:388-393): [start - max_track + 1, start] clipped at 0."""
    start = (pass_idx + 1) * frame_buffer_size - 1
    return max(start - max_frame_num_to_track + 1, 0), start


def pack_entry(entry: Dict[str, torch.Tensor]) -> torch.Tensor:
    """One flat uint8 buffer per cond-frame entry => a single collective per exchange."""
    parts = [entry[k].contiguous().view(torch.uint8).reshape(-1) for k in ENTRY_FIELDS]
    return torch.cat(parts)


def unpack_entry(buf: torch.Tensor, like: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out, off = {}, 0
    for k in ENTRY_FIELDS:
        n = like[k].numel() * like[k].element_size()
        out[k] = buf[off:off + n].view(like[k].dtype).reshape(like[k].shape).clone()
        off += n
    out["maskmem_pos_enc"] = None
    return out


def allgather_cond_entries(entry: Dict[str, torch.Tensor], group=None) -> List[Dict[str, torch.Tensor]]:
    """All-gather one cond-frame bank entry per rank (same object count on every rank).
    Returns the list of entries indexed by source rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [entry]
    flat = pack_entry(entry)
    bufs = [torch.empty_like(flat) for _ in range(dist.get_world_size(group))]
    dist.all_gather(bufs, flat, group=group)
    return [unpack_entry(b, entry) for b in bufs]


def install_cond_entry(predictor, inference_state, frame_idx: int, entry: Dict[str, torch.Tensor]) -> None:
    """Insert a cond-frame entry received from another rank into this rank's bank."""
    st = inference_state
    st["output_dict"]["cond_frame_outputs"][frame_idx] = entry
    st["consolidated_frame_inds"]["cond_frame_outputs"].add(frame_idx)
    predictor._add_output_per_object(st, frame_idx, entry, "cond_frame_outputs")
