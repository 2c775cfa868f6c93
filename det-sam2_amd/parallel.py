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


# ------------------------------------------------------------------------------------------------------------------
# One stream, passes sharded over ranks
# ------------------------------------------------------------------------------------------------------------------
def entry_template(B: int, device) -> Dict[str, torch.Tensor]:
    """Shapes/dtypes of one cond-frame bank entry for B objects (what a receiver allocates)."""
    return {"maskmem_features": torch.empty((B, 4096, 64), dtype=torch.bfloat16, device=device),
            "pred_masks": torch.empty((B, 1, 256, 256), dtype=torch.float32, device=device),
            "obj_ptr": torch.empty((B, 256), dtype=torch.float32, device=device),
            "object_score_logits": torch.empty((B, 1), dtype=torch.float32, device=device)}


def broadcast_cond_entries(payload, src: int, device, group=None):
    """The one data-path exchange of the pass-sharded stream: the owner of a pass broadcasts the object-id list and
    every conditioning-frame entry it created (or re-consolidated with more objects) in that pass.
    ``payload`` = {"obj_ids": [...], "entries": {frame_idx: entry}} on ``src``, ignored elsewhere.  Returns the
    payload on every rank.  Header by ``broadcast_object_list`` (a few bytes), tensors as one flat buffer each."""
    rank = dist.get_rank(group)
    hdr = [None]
    if rank == src:
        hdr = [(list(payload["obj_ids"]), [(int(t), int(e["obj_ptr"].shape[0])) for t, e in payload["entries"].items()])]
    dist.broadcast_object_list(hdr, src=src, group=group, device=torch.device(device) if str(device) != "cpu" else None)
    obj_ids, frames = hdr[0]
    out = {"obj_ids": obj_ids, "entries": {}}
    for t, B in frames:
        like = entry_template(B, device)
        if rank == src:
            buf = pack_entry({k: payload["entries"][t][k].to(device) for k in ENTRY_FIELDS})
        else:
            buf = torch.empty(sum(v.numel() * v.element_size() for v in like.values()), dtype=torch.uint8, device=device)
        dist.broadcast(buf, src=src, group=group)
        out["entries"][t] = payload["entries"][t] if rank == src else unpack_entry(buf, like)
    return out


def _make_sharded_cls():
    from .det_sam2_RT import VideoProcessor

    class ShardedVideoProcessor(VideoProcessor):
        """VideoProcessor whose propagate passes are sharded over ranks (pass k -> rank k mod world).

        Every rank is fed the SAME frame stream (ingest is 0.2 ms/frame) and keeps the same absolute frame indexing and
        eviction schedule; only the owner of a pass runs the detector, the prompts and the reverse propagation of
        that pass.  Afterwards it broadcasts the conditioning-frame entries it created - the only state a later pass
        on another rank needs, because non-conditioning memories and object pointers of a reverse pass are recomputed
        inside the pass (sam2_base.py:541-555,613).  Final masks of a frame = those of the LAST pass covering it
        (det_sam2_RT.py:396 overwrites), see ``merge_segments``.

        ``exchange(pass_idx, owner, payload) -> payload`` defaults to ``broadcast_cond_entries`` over
        ``torch.distributed``; tests inject an in-process mailbox.
        """

        def __init__(self, *a, rank=None, world_size=None, exchange=None, **kw):
            super().__init__(*a, **kw)
            init = dist.is_available() and dist.is_initialized()
            self.rank = rank if rank is not None else (dist.get_rank() if init else 0)
            self.world = world_size if world_size is not None else (dist.get_world_size() if init else 1)
            self._exchange = exchange or (lambda k, owner, payload: broadcast_cond_entries(payload, owner, self.predictor.device))
            self._pass_idx = 0
            self._shared_B = {}        # cond frame -> object count of the version every rank holds
            self.owned_passes = []

        def Detect_and_SAM2_inference(self, frame_idx):
            k, self._pass_idx = self._pass_idx, self._pass_idx + 1
            owner = pass_owner(k, self.world)
            past = self.inference_state["num_frames"] if self.inference_state else 0
            dets = self.detect_predict(self.frame_buffer, past) if owner == self.rank else {}
            self._ingest_buffer()
            st = self.inference_state
            payload = None
            if owner == self.rank:
                self.owned_passes.append(k)
                self._prompt_and_propagate(frame_idx, dets)
                cond = st["output_dict"]["cond_frame_outputs"]
                changed = {t: e for t, e in cond.items() if self._shared_B.get(t) != int(e["obj_ptr"].shape[0])}
                payload = {"obj_ids": list(st["obj_ids"]), "entries": changed}
            if self.world > 1:
                payload = self._exchange(k, owner, payload)
            if owner != self.rank:
                for oid in payload["obj_ids"]:           # same slot order as on the owner
                    if oid not in st["obj_id_to_idx"]:
                        self.predictor._new_slot(st, oid)
                for t, e in payload["entries"].items():
                    install_cond_entry(self.predictor, st, int(t), {**e, "maskmem_pos_enc": None})
            for t, e in (payload["entries"].items() if payload else []):
                self._shared_B[int(t)] = int(e["obj_ptr"].shape[0])
            st["tracking_has_started"] = True            # global fact: some rank has propagated (:899)
            self._release(frame_idx)
            self._log_pass(frame_idx)

    return ShardedVideoProcessor


def __getattr__(name):   # lazy: det_sam2_RT imports this module's siblings
    if name == "ShardedVideoProcessor":
        return _make_sharded_cls()
    raise AttributeError(name)


def merge_segments(per_rank_segments, frame_buffer_size: int, max_frame_num_to_track: int, num_passes: int,
                   world_size: int):
    """Final {frame: {obj_id: mask}} of a pass-sharded stream.  Pass k covers frames
    [(k+1)*buffer - track, (k+1)*buffer - 1]; of the passes covering frame t the LAST one wins (det_sam2_RT.py:396
    overwrites), i.e. the segments held by that pass' owner."""
    out = {}
    frames = sorted(set().union(*[set(s) for s in per_rank_segments]))
    for t in frames:
        last = min((t + max_frame_num_to_track) // frame_buffer_size - 1, num_passes - 1)
        out[t] = per_rank_segments[pass_owner(last, world_size)][t]
    return out
