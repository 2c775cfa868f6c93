"""Multi-GPU: ONE Det-SAM2 stream sharded over ranks by propagate pass, one process per GPU.

A Det-SAM2 pass k (the newest ``frame_buffer_size`` frames + the previous window, tracked in reverse) depends only on
(i) the images of its <= 2*buffer frames and (ii) the CONDITIONING-frame entries of the bank; non-conditioning memories
and object pointers are recomputed inside the pass (sam2_base.py:541-555,613).  And a conditioning entry itself depends
on nothing but its own frame and the object table: prompts on a never-tracked frame use no memory
(sam2_video_predictor.py:428, sam2_base.py:651-657).  So the stream is cut into ROUNDS of N consecutive passes,
pass k -> rank k mod N, and a round is

    1. every rank encodes the frames of ITS OWN buffer once and hands the pyramids to the owner of the next pass
       (ring shift over one xGMI link: buffer x 16 MiB) - a frame is encoded exactly once per stream although two passes
       (on two ranks) track it.  The shift is POSTED here and waited for only before step 5 (the received pyramids are
       first read by the propagation), so steps 2-4 run under the transfer;
    2. every rank runs the detector on its buffer; the detections are all-gathered as one tensor per rank ([1 + cap, 7]
       fp64: frame, xyxy, class, confidence; row 0 = count; cap = the round's largest row count, agreed on by a 4-byte
       count all-gather - no pickling, no fixed limit on the detections of a pass), so every
       rank derives the same object table for every pass of the round;
    3. every rank prompts + consolidates ITS conditioning frame(s) - ~50 ms, no dependence on other passes;
    4. ONE all-gather replicates the new conditioning entries (bf16 memory + masks + pointers + the frame's level-2
       feature: ~16 MiB per entry at 16 objects) - the only bank traffic, RCCL over xGMI;
    5. every rank replays the bank bookkeeping of the passes before its own (object-table growth, A17 re-consolidation,
       eviction - all deterministic and cheap), propagates its own pass with NO further communication (~2 s at
       hiera_l / 16 objects), then replays the rest of the round.

All ranks reach each collective after the same amount of work, so nobody waits on a propagating peer (the round-1
design broadcast after the owner's whole pass and serialised the ranks).  Latency: a round needs N buffers of frames, so
a live stream is delayed by N buffers; throughput scales with N.  Results are bit-identical to the sequential
``VideoProcessor`` (tests/test_hip_sharded.py), including a new class appearing mid-stream and frame eviction.

Only ``torch.distributed`` is used (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  The round logic is a
generator that yields its communication requests, so the same code runs under ``TorchDistComm`` and under the in-process
lock-step harness of the tests.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
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


# ------------------------------------------------------------------------------------------------------------------
# flat packing of bank entries (one buffer per collective)
# ------------------------------------------------------------------------------------------------------------------
def pack_entry(entry: Dict[str, torch.Tensor], fields=ENTRY_FIELDS) -> torch.Tensor:
    parts = [entry[k].contiguous().view(torch.uint8).reshape(-1) for k in fields]
    return torch.cat(parts)


def unpack_entry(buf: torch.Tensor, like: Dict[str, torch.Tensor], fields=ENTRY_FIELDS) -> Dict[str, torch.Tensor]:
    out, off = {}, 0
    for k in fields:
        n = like[k].numel() * like[k].element_size()
        out[k] = buf[off:off + n].view(like[k].dtype).reshape(like[k].shape).clone()
        off += n
    out["maskmem_pos_enc"] = None
    return out


def entry_template(B: int, device, tokens: int = 4096, mem_dim: int = 64, ptr_dim: int = 256, mask_side: int = 256,
                   feat_dim: int = 256) -> Dict[str, torch.Tensor]:
    """Shapes/dtypes of one exchanged conditioning entry for B objects: the bank entry + the frame's level-2 feature."""
    return {"maskmem_features": torch.empty((B, tokens, mem_dim), dtype=torch.bfloat16, device=device),
            "pred_masks": torch.empty((B, 1, mask_side, mask_side), dtype=torch.float32, device=device),
            "obj_ptr": torch.empty((B, ptr_dim), dtype=torch.float32, device=device),
            "object_score_logits": torch.empty((B, 1), dtype=torch.float32, device=device),
            "fpn2": torch.empty((tokens, feat_dim), dtype=torch.float32, device=device)}


XFIELDS = ENTRY_FIELDS + ("fpn2",)


def entry_nbytes(like) -> int:
    return sum(v.numel() * v.element_size() for v in like.values())


def allgather_cond_entries(entry: Dict[str, torch.Tensor], group=None) -> List[Dict[str, torch.Tensor]]:
    """All-gather one cond-frame bank entry per rank (same object count on every rank); list indexed by source rank."""
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
# communication back ends.  A round yields requests (op, payload); a back end answers them.
#   ("all_gather_tensor", tensor, same shape/rank)      -> [tensor of rank 0, ..., tensor of rank N-1]   (fixed size)
#   ("ring_shift_begin", (send tensors, recv tensors))  -> handle; rank r's send tensors travel to rank r+1's recv tensors
#   ("ring_shift_wait", handle)                         -> the recv tensors, filled
# ------------------------------------------------------------------------------------------------------------------
class TorchDistComm:
    """torch.distributed back end.  "nccl" (= RCCL on ROCm) moves device tensors directly over xGMI; under "gloo" (CPU
    tests, and single-GPU dry runs of the multi-rank code path) device tensors are staged through the host.  Small host
    tensors (detections, entry metadata) are moved to the device for RCCL and back."""

    def __init__(self, group=None, device=None):
        self.group, self.device = group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.host_staged = dist.get_backend(group) == "gloo"
        if not self.host_staged and self.device is None:   # RCCL moves device tensors only: small host tensors need a home
            if not torch.cuda.is_available():
                raise RuntimeError("TorchDistComm: backend 'nccl' (RCCL) needs a GPU; pass device= or use 'gloo'")
            self.device = torch.device("cuda", torch.cuda.current_device())

    def _wire(self, t):
        if self.host_staged:
            return t.cpu() if t.is_cuda else t
        return t if t.is_cuda else t.to(self.device)

    def execute(self, req):
        op, payload = req
        if op == "all_gather_tensor":
            src = self._wire(payload).contiguous()
            out = torch.empty(self.world * src.numel(), dtype=src.dtype, device=src.device)
            dist.all_gather_into_tensor(out, src.reshape(-1), group=self.group)     # one fixed-size collective
            out = out.to(payload.device).reshape((self.world,) + tuple(src.shape))
            return [out[i] for i in range(self.world)]
        if op == "ring_shift_begin":
            send, recv = payload
            nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
            s_ = [self._wire(t) for t in send]
            r_ = [torch.empty(t.shape, dtype=t.dtype) if (self.host_staged and t.is_cuda) else t for t in recv]
            ops = [dist.P2POp(dist.isend, t, nxt, group=self.group) for t in s_ if t.numel()]
            ops += [dist.P2POp(dist.irecv, t, prv, group=self.group) for t in r_ if t.numel()]
            works = dist.batch_isend_irecv(ops) if ops else []   # one ncclGroup: every rank sends and receives together
            return {"works": works, "recv": recv, "staged": r_, "keep": s_}
        if op == "ring_shift_wait":
            for w in payload["works"]:
                w.wait()
            for dst, got in zip(payload["recv"], payload["staged"]):
                if dst is not got and dst.numel():
                    dst.copy_(got)
            payload["keep"] = None
            return payload["recv"]
        raise ValueError(op)


def run_lockstep(generators):
    """In-process stand-in for N ranks: advance every rank's round generator to its next request, answer the N requests
    together, repeat.  Used by the tests and by single-GPU runs of the sharded driver (tests/test_hip_fullsize.py)."""
    gens = list(generators)
    reqs = [next(g, None) for g in gens]
    while any(r is not None for r in reqs):
        assert all(r is not None for r in reqs) and len({r[0] for r in reqs}) == 1, "ranks diverged"
        op, n = reqs[0][0], len(gens)
        if op == "all_gather_tensor":
            assert len({(tuple(r[1].shape), r[1].dtype) for r in reqs}) == 1, "all_gather_tensor: ranks disagree on the shape"
            res = [[r[1].clone() for r in reqs] for _ in range(n)]
        elif op == "ring_shift_begin":
            res = []
            for i in range(n):
                send = reqs[(i - 1) % n][1][0]
                recv = reqs[i][1][1]
                for dst, src in zip(recv, send):
                    assert dst.shape == src.shape and dst.dtype == src.dtype, (dst.shape, src.shape)
                    if dst.numel():
                        dst.copy_(src)
                res.append({"recv": recv})
        elif op == "ring_shift_wait":
            res = [r[1]["recv"] for r in reqs]
        else:
            raise ValueError(op)
        nxt = []
        for g, r in zip(gens, res):
            try:
                nxt.append(g.send(r))
            except StopIteration:
                nxt.append(None)
        reqs = nxt


def drive_lockstep(vps, frames):
    """Feed one frame stream to N in-process ShardedVideoProcessors (ranks 0..N-1) and run their rounds in lock step."""
    b, n = vps[0].frame_buffer_size, vps[0].world
    assert [v.rank for v in vps] == list(range(n))
    idx = -1
    for idx, fr in enumerate(frames):
        for v in vps:
            v.frame_buffer.append(fr)
        if len(vps[0].frame_buffer) >= b * n:
            run_lockstep([v.round_generator(v.pre_frames + idx) for v in vps])
            for v in vps:
                v.frame_buffer.clear()
    if vps[0].frame_buffer:
        run_lockstep([v.round_generator(v.pre_frames + idx) for v in vps])
        for v in vps:
            v.frame_buffer.clear()


# ------------------------------------------------------------------------------------------------------------------
# One stream, passes sharded over ranks
# ------------------------------------------------------------------------------------------------------------------
def _make_sharded_cls():
    from .det_sam2_RT import VideoProcessor

    class ShardedVideoProcessor(VideoProcessor):
        """VideoProcessor whose propagate passes are sharded over ranks in rounds of ``world_size`` passes (module
        docstring).  Every rank is fed the SAME frame stream through ``process_frame`` / ``run`` (ingest is 0.2 ms per
        frame) and keeps the same absolute frame indexing, object table and eviction schedule; it runs the detector,
        the prompts and the reverse propagation only for its own passes.  Final masks of a frame = those of the LAST
        pass covering it (det_sam2_RT.py:396 overwrites), see ``merge_segments``.

        ``comm``: object with ``execute(request)`` (default ``TorchDistComm``).  With ``comm=None`` and no initialised
        process group the rounds are driven from outside through ``round_generator()`` (``run_lockstep``)."""

        def __init__(self, *a, rank=None, world_size=None, comm=None, handoff_features=True, **kw):
            super().__init__(*a, **kw)
            init = dist.is_available() and dist.is_initialized()
            self.rank = rank if rank is not None else (dist.get_rank() if init else 0)
            self.world = world_size if world_size is not None else (dist.get_world_size() if init else 1)
            self.comm = comm if comm is not None else (TorchDistComm(device=self.predictor.device) if init else None)
            self.handoff = handoff_features
            self._round_frames = []          # frames of the current round, buffer by buffer
            self._next_pass = 0
            self._prev_buffer_feats = {}     # rank 0: features received at the end of round R for its pass of round R+1
            self._host_frames = {}           # absolute index -> host frame (references) of the last rounds
            self.owned_passes = []
            self.comm_log = []               # (round, op, bytes) for the bench / tests
            self.profile_rounds = False      # bench: wall-time split of every round (device-synchronised phase boundaries)
            self.round_times = []            # [{"round", "ingest", "encode", "detect", "prompt", "gather", "ring_wait", "propagate"}] seconds
            assert self.detect_interval == -1 or self.detect_interval > 0

        # -------------------------------------------------------------- frame intake: a round = world x buffer frames
        def process_frame(self, frame_idx, frame):
            self.frame_buffer.append(frame)
            if len(self.frame_buffer) >= self.frame_buffer_size * self.world:
                self.Detect_and_SAM2_inference(frame_idx)
                self.frame_buffer.clear()
            return self.inference_state

        def Detect_and_SAM2_inference(self, frame_idx):
            """One ROUND over the buffered frames (up to world passes; the last round of a stream may be shorter)."""
            gen = self.round_generator(frame_idx)
            if self.world == 1:
                for _ in gen:        # a 1-rank round has nobody to talk to
                    raise AssertionError("unexpected communication request at world size 1")
                return
            if self.comm is None:
                raise RuntimeError("ShardedVideoProcessor: no communication back end (initialise torch.distributed, pass "
                                   "comm=, or drive round_generator() with run_lockstep)")
            try:
                req = next(gen)
                while True:
                    req = gen.send(self.comm.execute(req))
            except StopIteration:
                pass

        # -------------------------------------------------------------- helpers
        def _ingest_round(self, first_abs, lo, hi):
            """Ingest (H2D + resize/normalise) only the frames [lo, hi] of the round buffer - this rank's own buffer; the
            other frames of the round are never needed as images here (their pyramids arrive by hand-off, the level-2
            features of peer conditioning frames with the entries).  Frame numbering still advances by the whole round.
            The host frames of the last two rounds stay reachable for the rare on-demand ingest (partial final window)."""
            p, n_round = self.predictor, len(self.frame_buffer)
            for i, fr in enumerate(self.frame_buffer):
                self._host_frames[first_abs + i] = fr
            for t in [t for t in self._host_frames if t < first_abs - n_round - 2 * self.frame_buffer_size]:
                del self._host_frames[t]
            own = self.frame_buffer[lo - first_abs: hi - first_abs + 1] if hi >= lo else []
            idx = list(range(lo, hi + 1))
            if self.inference_state is None:
                seed = own if own else self.frame_buffer[:1]
                st = p.init_state(video_path=seed, warm_up_first_frame=False)
                st["images_idx"] = idx if own else [first_abs]
                st["num_frames"] = n_round
                self.inference_state = st
            else:
                p.append_sparse_frames(self.inference_state, own, idx, advance=n_round)
            self.inference_state["_frame_source"] = self._host_frames.get

        def _pass_abs_range(self, first_abs, j_local, n_frames_round):
            """Absolute frame indices of the NEW frames of the round's j-th pass."""
            b = self.frame_buffer_size
            lo = first_abs + j_local * b
            return lo, min(lo + b, first_abs + n_frames_round) - 1

        def _new_ids(self, dets):
            """Class ids of a pass' detections that become tracked objects, in the order the sequential driver meets them
            (Detect_2_SAM2_Prompt, det_sam2_RT.py:267-316)."""
            out = []
            for key in dets or {}:
                for d in dets[key]:
                    c = int(np.asarray(d["class"]).reshape(-1)[0])
                    if c not in self.skip_classes and c not in out:
                        out.append(c)
            return out

        def _apply_special(self, dets):
            """The special-class bookkeeping of detect_predict (det_sam2_RT.py:248-260) for a pass detected elsewhere."""
            for key in dets or {}:
                ds = dets[key]
                if not self.special_classes_detection:
                    self.special_classes_count = 0
                cls = [int(np.asarray(d["class"]).reshape(-1)[0]) for d in ds]
                n_special = sum(1 for c in cls if c == self.special_classes)
                if n_special > self.special_classes_count:
                    self.special_classes_detection = [d["coordinates"] for d, c in zip(ds, cls) if c == self.special_classes]
                    self.special_classes_count = n_special

        def _grow_table(self, ids):
            st = self.inference_state
            for oid in ids:
                if oid not in st["obj_id_to_idx"]:
                    self.predictor._new_slot(st, oid)

        def _sync_bank_batch(self):
            """A17 (sam2_video_predictor.py:281-310) made explicit: every live conditioning entry among the latest
            max_update_length_for_new_obj_id (+ all preload ones) whose batch is smaller than the object table is
            re-consolidated: placeholders for the new objects, memory encoder re-run.  Deterministic per object, so
            every rank obtains the entries the sequential driver would hold."""
            st, p = self.inference_state, self.predictor
            B = p._get_obj_num(st)
            od = st["output_dict"]
            inds = sorted(od["cond_frame_outputs"].keys())
            mx = st["max_update_length_for_new_obj_id"]
            if mx > 0:
                inds = inds[-mx:]
            for t in st["preloading_memory_cond_frame_idx"] or []:
                if t not in inds:
                    inds.append(t)
            for t in inds:
                if od["cond_frame_outputs"][t]["obj_ptr"].shape[0] < B:
                    cons = p._consolidate(st, t, True, True)
                    od["cond_frame_outputs"][t] = cons
                    p._add_output_per_object(st, t, cons, "cond_frame_outputs")

        def _canonical_cond_order(self):
            """The sequential driver inserts conditioning entries in stream order (after the preload ones), and the bank
            is assembled in dict order: keep that order here whatever order own / received entries were installed in, so
            the keys reach the attention kernel in the same sequence (bit-identical accumulation)."""
            st = self.inference_state
            cond = st["output_dict"]["cond_frame_outputs"]
            pre = [t for t in (st["preloading_memory_cond_frame_idx"] or []) if t in cond]
            rest = sorted(t for t in cond if t not in pre)
            if list(cond) != pre + rest:
                st["output_dict"]["cond_frame_outputs"] = {t: cond[t] for t in pre + rest}

        def _cond_frames_of(self, lo, hi):
            if self.detect_interval == -1:
                return []
            return [t for t in range(lo, hi + 1) if t % self.detect_interval == 0]

        def _like(self, B):
            return entry_template(B, self.predictor.device, **self.predictor.entry_dims())

        def _tick(self, rec, name, t0):
            """Close phase `name` of the round's time record (only when profiling: it synchronises the device)."""
            if not self.profile_rounds:
                return t0
            import time
            if self.predictor.device.type == "cuda":
                torch.cuda.synchronize(self.predictor.device)
            t1 = time.perf_counter()
            rec[name] = rec.get(name, 0.0) + (t1 - t0)
            return t1

        # -------------------------------------------------------------- the round
        def round_generator(self, frame_idx):
            """Generator over the communication requests of one round (module docstring, steps 1-5)."""
            p, b, N, r = self.predictor, self.frame_buffer_size, self.world, self.rank
            n_round = len(self.frame_buffer)
            n_pass = -(-n_round // b)
            k0 = self._next_pass
            self._next_pass += n_pass
            round_idx = k0 // N
            past = self.inference_state["num_frames"] if self.inference_state else 0
            first_abs = past                                   # absolute index of the round's first frame
            mine = r if r < n_pass else None                   # local index of my pass in this round
            import time
            rec, tp = {"round": round_idx}, time.perf_counter()
            # ---- frame numbering and eviction are the same on every rank; the images a rank holds are its own buffers
            if N > 1 and self.handoff:
                lo_, hi_ = self._pass_abs_range(first_abs, mine, n_round) if mine is not None else (0, -1)
                self._ingest_round(first_abs, lo_, hi_)
            else:
                self._ingest_buffer()
            st = self.inference_state
            d = p.device
            tp = self._tick(rec, "ingest", tp)
            # ---- 1. encode my buffer once, POST the hand-off of the pyramids to the owner of the next pass
            ring = None
            if self.handoff and N > 1:
                lo, hi = self._pass_abs_range(first_abs, mine, n_round) if mine is not None else (0, -1)
                my_frames = list(range(lo, hi + 1))
                st["_feature_cache_cap"] = 3 * b + p.encode_batch + 2
                feats = p.encode_frames(st, my_frames) if my_frames else []
                tp = self._tick(rec, "encode", tp)
                # what I receive: the buffer of local pass (r-1) of this round; rank 0 receives the round's last buffer
                src_local = (r - 1) % N
                n_recv, rlo = 0, 0
                if src_local < n_pass:
                    rlo, rhi = self._pass_abs_range(first_abs, src_local, n_round)
                    n_recv = rhi - rlo + 1
                shapes = p.feature_shapes()
                send = [torch.stack([f[i] for f in feats]) if feats else torch.empty((0,) + s, device=d) for i, s in enumerate(shapes)]
                recv = [torch.empty((n_recv,) + s, dtype=torch.float32, device=d) for s in shapes]
                if d.type == "cuda":      # the collective runs on the back end's own stream: the operands must be complete
                    torch.cuda.current_stream(d).synchronize()
                ring = yield ("ring_shift_begin", (send, recv))
                self.comm_log.append((round_idx, "ring_shift", sum(t.numel() * 4 for t in send)))
            # ---- 2. detections of my buffer -> everybody (one fixed-size tensor per rank)
            dets = {}
            if mine is not None:
                lo, hi = self._pass_abs_range(first_abs, mine, n_round)
                keep = (self.special_classes_detection, self.special_classes_count)
                dets = self.detect_predict_range(lo, hi, first_abs)
                self.special_classes_detection, self.special_classes_count = keep     # applied in pass order in step 5
            tp = self._tick(rec, "detect", tp)
            if N > 1:
                # rows on the wire = the round's largest pass (counts first, 4 bytes per rank): no fixed cap on the number of
                # detections of a pass (detect_interval 1, buffer 30, 16+ objects is 480 rows)
                n_rows = yield ("all_gather_tensor", torch.tensor([dets_count(dets)], dtype=torch.int32))
                cap = max(1, max(int(x[0]) for x in n_rows))
                all_dets = yield ("all_gather_tensor", dets_to_tensor(dets, cap))
                self.comm_log.append((round_idx, "all_gather_dets", (4 + (1 + cap) * 7 * 8) * N))
                all_dets = [dets_from_tensor(x) for x in all_dets][:n_pass]
            else:
                all_dets = [dets]
            if mine is not None:
                all_dets[mine] = dets                    # my own detections, not their wire copy
            tp = self._tick(rec, "gather", tp)
            tables, cur = [], list(st["obj_ids"])
            for j in range(n_pass):
                for c in self._new_ids(all_dets[j]):
                    if c not in cur:
                        cur.append(c)
                tables.append(list(cur))
            # ---- 3. my conditioning entries (prompts + consolidation; the object table is the one of MY pass)
            my_entries = {}
            if mine is not None:
                self._grow_table(tables[mine])
                self.inference_state = self.Detect_2_SAM2_Prompt(dets)
                p.propagate_in_video_preflight(st)
                lo, hi = self._pass_abs_range(first_abs, mine, n_round)
                for t in self._cond_frames_of(lo, hi):
                    e = st["output_dict"]["cond_frame_outputs"].get(t)
                    if e is not None and e["maskmem_features"] is not None:
                        my_entries[t] = dict(e, fpn2=p._get_image_feature(st, t)[2])
            tp = self._tick(rec, "prompt", tp)
            # ---- 4. one all-gather of the round's new conditioning entries (padded to the largest pass payload); what each
            #         pass contributes (frame, batch) travels first as a fixed-size int32 tensor
            meta = {t: int(e["obj_ptr"].shape[0]) for t, e in my_entries.items()}
            if N > 1:
                # a pass announces at most one entry per frame it holds detections for - known to every rank from step 2
                cond_cap = max(1, max(len(x) for x in all_dets))
                metas = yield ("all_gather_tensor", meta_to_tensor(meta, cond_cap))
                metas = [meta_from_tensor(x) for x in metas][:n_pass]
            else:
                metas = [meta]
            if any(metas) and N > 1:
                sizes = [sum(entry_nbytes(self._like(Bj)) for Bj in m.values()) for m in metas]
                flat = torch.zeros(max(sizes), dtype=torch.uint8, device=d)
                off = 0
                for t in sorted(my_entries):
                    buf = pack_entry(my_entries[t], XFIELDS)
                    flat[off:off + buf.numel()] = buf
                    off += buf.numel()
                if d.type == "cuda":
                    torch.cuda.current_stream(d).synchronize()
                gathered = yield ("all_gather_tensor", flat)
                self.comm_log.append((round_idx, "all_gather_entries", int(flat.numel()) * N))
            else:
                gathered = [None] * N
            incoming = {}
            for j, m in enumerate(metas):
                off = 0
                for t in sorted(m):
                    like = self._like(m[t])
                    if j != mine:
                        incoming.setdefault(j, {})[t] = unpack_entry(gathered[j][off:off + entry_nbytes(like)], like, XFIELDS)
                    off += entry_nbytes(like)
            tp = self._tick(rec, "gather", tp)
            # ---- the pyramids posted in step 1 are first read by the propagation: wait for them here
            if ring is not None:
                recv = yield ("ring_shift_wait", ring)
                got = {rlo + i: tuple(t[i] for t in recv) for i in range(n_recv)} if n_recv else {}
                if r == 0:   # what arrives now belongs to my pass of the NEXT round; this round uses the previous arrival
                    use, self._prev_buffer_feats = self._prev_buffer_feats, got
                else:
                    use = got
                for t, f in use.items():
                    st["cached_features"][t] = f
                tp = self._tick(rec, "ring_wait", tp)
            # ---- 5. replay the passes before mine, run mine, replay the rest
            for j in range(n_pass):
                f_idx = min(first_abs + (j + 1) * b, first_abs + n_round) - 1       # newest frame of pass j
                self._apply_special(all_dets[j])
                self._grow_table(tables[j])
                for t, e in (incoming.get(j) or {}).items():
                    f2 = e.pop("fpn2")
                    if t not in st["cached_features"]:       # level-2 feature of a peer's conditioning frame (for A17)
                        st["cached_features"][t] = (None, None, f2)
                        st.setdefault("_pinned_features", set()).add(t)
                    install_cond_entry(p, st, t, e)
                self._canonical_cond_order()
                self._sync_bank_batch()
                st["tracking_has_started"] = st["tracking_has_started"] or bool(st["output_dict"]["cond_frame_outputs"])
                if j == mine:
                    self.owned_passes.append(k0 + j)
                    self._propagate_owned(f_idx)
                self._release(f_idx)
                self._log_pass(f_idx)
            tp = self._tick(rec, "propagate", tp)
            if self.profile_rounds:
                self.round_times.append(rec)

        def _propagate_owned(self, frame_idx):
            """The reverse propagation + host copy of an owned pass (prompts were issued in step 3)."""
            self._yielded, packed = [], []
            from .det_sam2_RT import PackedMasks
            st = self.inference_state
            # the whole round is ingested up front, but the bank logic reads num_frames (number of object pointers and
            # their temporal normalisation, sam2_base.py:591-633): during my pass it is what the sequential driver has
            # at that point - everything up to the pass' newest frame
            total, st["num_frames"] = st["num_frames"], frame_idx + 1
            try:
                for t, obj_ids, bits in self.predictor.propagate_in_video(
                        st, start_frame_idx=frame_idx, max_frame_num_to_track=self.max_frame_num_to_track, reverse=True, output="packed"):
                    self._yielded.append(t)
                    if t >= self.pre_frames:
                        packed.append((t, list(obj_ids), bits))
            finally:
                st["num_frames"] = total
            if packed:
                host = torch.stack([x for _, _, x in packed]).cpu().numpy()
                for (t, ids, _), pb in zip(packed, host):
                    self.video_segments[t] = PackedMasks(pb, ids, st["video_width"])

        def detect_predict_range(self, lo, hi, first_abs):
            """detect_predict (det_sam2_RT.py:201-265) restricted to the absolute frames [lo, hi] of the round buffer."""
            return self.detect_predict(self.frame_buffer[lo - first_abs: hi - first_abs + 1], lo)

    return ShardedVideoProcessor


def dets_count(dets) -> int:
    return sum(len(v) for v in (dets or {}).values())


def dets_to_tensor(dets, cap: int) -> torch.Tensor:
    """{frame: [detection dict]} (detect_predict's output, det_sam2_RT.py:228-244) -> fp64 [1 + cap, 7] (``cap`` = the
    largest row count of the round, agreed on by a count all-gather just before, so every rank sends the same shape): row 0 =
    [count, 0...]; row i = [frame, x0, y0, x1, y1, class, confidence] (the dict keys are "frame_<i>", det_sam2_RT.py:224).  fp32 boxes / classes / confidences and frame
    indices are exact in fp64; detection order (which decides object-table order) is the row order."""
    rows = [(float(str(k).rsplit("_", 1)[-1]), *np.asarray(d["coordinates"], np.float32).reshape(4).astype(np.float64).tolist(),
             float(np.asarray(d["class"], np.float32).reshape(-1)[0]), float(np.asarray(d["confidence"], np.float32).reshape(-1)[0]))
            for k, v in (dets or {}).items() for d in v]
    if len(rows) > cap:
        raise RuntimeError(f"{len(rows)} detections in one pass exceed the agreed wire size {cap}")
    out = torch.zeros((1 + cap, 7), dtype=torch.float64)
    out[0, 0] = len(rows)
    if rows:
        out[1:1 + len(rows)] = torch.tensor(rows, dtype=torch.float64)
    return out


def dets_from_tensor(t: torch.Tensor):
    t = t.cpu()
    out = {}
    for row in t[1:1 + int(t[0, 0])].tolist():
        out.setdefault(f"frame_{int(row[0])}", []).append(
            {"coordinates": np.asarray(row[1:5], np.float32), "class": np.array([row[5]], np.float32),
             "confidence": np.array([row[6]], np.float32)})
    return out


def meta_to_tensor(meta, cap: int) -> torch.Tensor:
    """{cond frame: batch size} -> int32 [1 + cap, 2] (row 0 = count; ``cap`` = the largest number of detection frames of
    a pass of the round, which every rank knows after the detection all-gather)."""
    if len(meta) > cap:
        raise RuntimeError(f"{len(meta)} conditioning frames in one pass exceed the agreed wire size {cap}")
    out = torch.zeros((1 + cap, 2), dtype=torch.int32)
    out[0, 0] = len(meta)
    for i, t in enumerate(sorted(meta)):
        out[1 + i, 0], out[1 + i, 1] = int(t), int(meta[t])
    return out


def meta_from_tensor(t: torch.Tensor):
    t = t.cpu()
    return {int(a): int(b) for a, b in t[1:1 + int(t[0, 0])].tolist()}


_SHARDED_CLS = None


def __getattr__(name):   # lazy (det_sam2_RT imports this module's siblings); ONE class object, so isinstance works
    global _SHARDED_CLS
    if name == "ShardedVideoProcessor":
        if _SHARDED_CLS is None:
            _SHARDED_CLS = _make_sharded_cls()
        return _SHARDED_CLS
    raise AttributeError(name)


def merge_segments(per_rank_segments, frame_buffer_size: int, max_frame_num_to_track: int, num_passes: int,
                   world_size: int, num_frames: int = None):
    """Final {frame: {obj_id: mask}} of a pass-sharded stream.  Pass k starts at its newest frame
    s_k = min((k+1)*buffer, num_frames) - 1 (the last pass of a stream may be a partial buffer) and covers
    [s_k - track + 1, s_k]; of the passes covering frame t the LAST one wins (det_sam2_RT.py:396 overwrites), i.e. the
    segments held by that pass' owner."""
    out = {}
    frames = sorted(set().union(*[set(s) for s in per_rank_segments]))
    n = num_frames if num_frames is not None else num_passes * frame_buffer_size
    starts = [min((k + 1) * frame_buffer_size, n) - 1 for k in range(num_passes)]
    for t in frames:
        last = max(k for k in range(num_passes) if starts[k] - max_frame_num_to_track + 1 <= t <= starts[k])
        out[t] = per_rank_segments[pass_owner(last, world_size)][t]
    return out
