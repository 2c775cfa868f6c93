"""SAM2VideoPredictor for MI355X: the reference's predictor API and inference-state schema
(sam2/sam2_video_predictor.py) over the HIP stages of ``HipSam2``.

Same public methods, argument meaning and error behaviour as the reference so that
``VideoProcessor`` / ``DetSAM2Pipeline`` callers stay drop-in:
``init_state, update_state, init_preloading_state, add_new_points_or_box, propagate_in_video
(generator), release_old_frames, reset_state``.

What is re-designed (results unchanged):
* everything stays in HBM: frames (fp16), per-frame pyramid features, the memory bank (bf16) - no
  host offload, no per-object ``.cpu()``, no ``empty_cache()``;
* image features are cached for every retained frame, so the second visit of a frame in the next
  pass (README.md:161 of the reference) does not re-run the encoder;
* the memory bank is indexed, not concatenated on the host: one gather kernel per tracked frame;
* tensors are token-major: ``pred_masks`` [B,1,256,256] as in the reference, ``maskmem_features`` bf16
  [B,4096,64] (reference: [B,64,64,64]); ``maskmem_pos_enc`` is a model constant and is stored as None.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch

from .config import resolve_config
from .hip_model import HipSam2

NO_OBJ_SCORE = -1024.0  # sam2_base.py:21


class SAM2VideoPredictor:
    def __init__(self, cfg, state_dict, device="cuda:0", max_batch=16, fill_hole_area=0, hip=None):
        """``hip``: an object with HipSam2's stage interface (tests of the host logic inject a CPU stand-in; the product
        path always builds HipSam2, which raises without the HIP library or a GPU)."""
        self.cfg = resolve_config(cfg)
        # sam2_video_predictor.py:26,37: class default 0; build_sam2_video_predictor passes 8 (build_sam.py:134).
        # The reference applies it only when its CUDA extension is present (misc.py:389-391: the CPU reference, and
        # therefore the committed goldens, skip it) - here it is always applied when > 0 (HIP kernel, A14).
        self.fill_hole_area = int(fill_hole_area)
        self.hip = hip if hip is not None else HipSam2(self.cfg, state_dict, device, max_batch)
        self.device = self.hip.device
        self.image_size = self.cfg.image_size
        self.hidden_dim, self.mem_dim, self.num_maskmem = self.cfg.d_model, self.cfg.mem_dim, self.cfg.num_maskmem
        self.trace = None          # optional list of bank-selection traces (tests)
        self.stats = {"encoder_runs": 0, "encoder_launches": 0, "tracked_frames": 0}
        # frames encoded per image-encoder launch: the driver hands frames over 30 at a time, and one frame's
        # Hiera stage-3/4 GEMMs (4096 / 1024 tokens) cannot fill 256 CUs.  Same results as one-by-one.
        # Upper bound 16 (the C-ABI's limit); the actual batch splits the frames still to encode evenly (20 -> 10 + 10,
        # 30 -> 15 + 15): at 16 frames every Hiera stage-3 GEMM of hiera_l is a whole number of 256-CU rounds.
        self.encode_batch = int(os.environ.get("DS2_ENCODE_BATCH", "16"))
        # The next encoder batch is launched AHEAD of need on a second HIP stream, through a VIEW of the model (the same weights,
        # its own workspace arena: ds2_model_create_view), so that its large GEMMs fill the CUs the small kernels of the tracking chain (SAM
        # heads, memory encoder) leave idle: +4.5 % frames/s at hiera_l / 16 objects.  Same kernels, same results.
        # Overlapped kernels share CUs, so every per-kernel duration (and the bench's roofline fraction) reads ~5 %
        # worse than in isolation; DS2_ASYNC_ENCODE=0 switches it off.
        self.async_encode = hip is None and os.environ.get("DS2_ASYNC_ENCODE", "1") not in ("", "0")
        self.async_lookahead = 12
        self._hip_enc, self._enc_stream = None, None

    # ------------------------------------------------------------------ frame ingest (A3)
    def _load_frames(self, video_path):
        if isinstance(video_path, np.ndarray) and video_path.ndim == 3:
            video_path = [video_path]
        if isinstance(video_path, torch.Tensor):       # already-resident uint8 [n,H,W,3] frames
            u8 = video_path
        elif isinstance(video_path, (list, tuple)) and all(isinstance(p, np.ndarray) for p in video_path):
            u8 = torch.from_numpy(np.ascontiguousarray(np.stack(video_path)))
        elif isinstance(video_path, np.ndarray) and video_path.ndim == 4:
            u8 = torch.from_numpy(np.ascontiguousarray(video_path))
        else:
            raise NotImplementedError(
                "only in-memory RGB uint8 frames (ndarray / list of ndarray / uint8 tensor) are supported; "
                "JPEG folders and video files are outside the hot path (misc.py:292-303)")
        h, w = int(u8.shape[1]), int(u8.shape[2])
        images = self.hip.ingest(u8.to(self.device, non_blocking=True).contiguous())
        return images, h, w

    @torch.inference_mode()
    def init_state(self, video_path, offload_video_to_cpu=True, offload_state_to_cpu=False, async_loading_frames=False,
                   warm_up_first_frame=True):
        """init_state (sam2_video_predictor.py:44-120).  The offload flags are accepted for signature
        compatibility; state always lives in HBM here.  ``warm_up_first_frame=False`` skips the reference's encoder
        warm-up on frame 0 (:118; ranks of a sharded stream that do not own the first pass never need that feature)."""
        images, vh, vw = self._load_frames(video_path)
        st = {}
        st["images"] = images
        st["num_frames"] = len(images)
        st["images_idx"] = list(range(len(images)))
        st["offload_video_to_cpu"] = False
        st["offload_state_to_cpu"] = False
        st["video_height"], st["video_width"] = vh, vw
        st["device"] = self.device
        st["storage_device"] = self.device
        st["point_inputs_per_obj"] = {}
        st["mask_inputs_per_obj"] = {}
        st["cached_features"] = {}
        st["_pending_features"] = {}
        st["constants"] = {}
        st["obj_id_to_idx"] = OrderedDict()
        st["obj_idx_to_id"] = OrderedDict()
        st["obj_ids"] = []
        st["output_dict"] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        st["output_dict_per_obj"] = {}
        st["temp_output_dict_per_obj"] = {}
        st["consolidated_frame_inds"] = {"cond_frame_outputs": set(), "non_cond_frame_outputs": set()}
        st["tracking_has_started"] = False
        st["frames_already_tracked"] = {}
        st["preloading_memory_cond_frame_idx"] = None
        st["preloading_memory_non_cond_frames_idx"] = None
        st["max_update_length_for_new_obj_id"] = 100
        if warm_up_first_frame:
            self._get_image_feature(st, 0)
        return st

    @torch.inference_mode()
    def update_state(self, video_path, inference_state, async_loading_frames=False):
        """update_state (sam2_video_predictor.py:160-204)."""
        st = inference_state
        new, vh, vw = self._load_frames(video_path)
        assert vh == st["video_height"] and vw == st["video_width"], "new frames must match the video size"
        # a preloaded DS2BANK state holds no frames: numbering continues after the bank (num_frames), as it does in the
        # reference where the bank's frames are still in `images`
        last = st["images_idx"][-1] if st["images_idx"] else st["num_frames"] - 1
        st["images_idx"].extend(range(last + 1, last + 1 + len(new)))
        st["images"] = torch.cat((st["images"], new), dim=0) if len(st["images"]) else new
        st["num_frames"] += len(new)
        return st

    def append_sparse_frames(self, inference_state, frames, abs_indices, advance=0):
        """Ingest ``frames`` under the given ABSOLUTE frame indices (not necessarily contiguous with what is retained) and
        advance ``num_frames`` by ``advance``.  For drivers that hold only part of a stream on this GPU (the pass-sharded
        driver ingests its own buffer; features of the other frames arrive from peer ranks)."""
        st = inference_state
        if len(frames):
            new, vh, vw = self._load_frames(list(frames))
            assert vh == st["video_height"] and vw == st["video_width"], "new frames must match the video size"
            st["images_idx"].extend(int(t) for t in abs_indices)
            st["images"] = torch.cat((st["images"], new), dim=0) if len(st["images"]) else new
        st["num_frames"] += int(advance)
        return st

    def init_preloading_state(self, inference_state, offload_video_to_cpu=True, offload_state_to_cpu=True):
        """init_preloading_state (sam2_video_predictor.py:123-156): in the reference this moves the preload bank to the
        storage device; here the bank (loaded by bank_io.load_bank: DS2BANK file or a reference pickle, both already
        converted to the token-major layout) is moved into HBM and the per-object views are rebuilt."""
        st = inference_state
        st["storage_device"] = st["device"] = self.device
        st["images"] = st["images"].to(self.device)
        B = len(st["obj_ids"])
        for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
            for t, out in st["output_dict"][key].items():
                f = out.get("maskmem_features")
                if f is not None and f.dim() == 4:       # a reference-layout state handed over directly (not via load_bank)
                    f = f.flatten(2).transpose(1, 2).contiguous().to(torch.bfloat16)
                if f is not None and tuple(f.shape[1:]) != (4096, 64):
                    raise ValueError(f"preload bank entry of frame {t}: maskmem_features has shape {tuple(f.shape)}, expected "
                                     "token-major [B,4096,64] or the reference's [B,64,64,64]")
                out["maskmem_features"] = f
                out["maskmem_pos_enc"] = None
                for k in ("maskmem_features", "pred_masks", "obj_ptr", "object_score_logits"):
                    if out.get(k) is not None:
                        out[k] = out[k].to(self.device).contiguous()
        st["output_dict_per_obj"] = {i: {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}} for i in range(B)}
        for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
            for t, out in st["output_dict"][key].items():
                self._add_output_per_object(st, t, out, key)
        # level-2 features of the bank's conditioning frames (DS2BANK carries them instead of frames): pinned in the
        # feature cache for the online new-object re-consolidation (A17), which only runs the memory encoder on them
        st["cached_features"] = {}
        st["_pending_features"] = {}
        st["_pinned_features"] = set()
        for t, f2 in (st.get("preload_fpn2") or {}).items():
            f2 = f2.to(self.device).contiguous()
            st["preload_fpn2"][t] = f2
            st["cached_features"][t] = (None, None, f2)
            st["_pinned_features"].add(t)

    # ------------------------------------------------------------------ features (A4/A5)
    def _get_image_feature(self, st, frame_idx):
        """_get_image_feature (sam2_video_predictor.py:1174-1212) with a whole-window cache and batched encoding:
        on a miss, the next not-yet-encoded frames of the current propagation order ride along in one launch."""
        cache = st["cached_features"]
        f = cache.get(frame_idx)
        pend = st.get("_pending_features")
        if f is None and pend and frame_idx in pend:       # encoded ahead on the side stream: wait for its event, adopt it
            f, ev = pend.pop(frame_idx)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for x in f:
                x.record_stream(cur)
            cache[frame_idx] = f
            self._trim_feature_cache(st, keep={frame_idx})
        if f is None and frame_idx not in st["images_idx"] and st.get("_frame_source") is not None:
            fr = st["_frame_source"](frame_idx)          # a frame this GPU did not ingest (sharded driver): fetch it now
            if fr is not None:
                self.append_sparse_frames(st, [fr], [frame_idx])
        if f is None and frame_idx not in st["images_idx"]:
            raise RuntimeError(f"frame {frame_idx} has neither a retained image nor cached features (released, or a preload-"
                               "bank frame stored without its level-2 feature): it cannot be encoded")
        if f is None:
            todo = [frame_idx]
            order = st.get("_encode_order")
            if self.encode_batch > 1 and order is not None and frame_idx in order:
                have = set(st["images_idx"])
                rest = [t for t in order[order.index(frame_idx) + 1:] if t not in cache and t in have]
                n_rem = 1 + len(rest)
                n_batches = -(-n_rem // self.encode_batch)
                todo += rest[: -(-n_rem // n_batches) - 1]           # even split of what is left over the batches it needs
            if len(todo) == 1:
                feats = [self.hip.image_encoder(st["images"][st["images_idx"].index(frame_idx)])]
            else:
                pos = torch.tensor([st["images_idx"].index(t) for t in todo], device=st["images"].device)
                feats = self.hip.image_encoder_batch(st["images"].index_select(0, pos))
            for t, ft in zip(todo, feats):
                cache[t] = ft
            self._trim_feature_cache(st, keep=set(todo))
            self.stats["encoder_runs"] += len(todo)
            self.stats["encoder_launches"] += 1
            f = cache[frame_idx]
        if self.async_encode:
            self._prefetch_features(st, frame_idx)
        return f

    def _prefetch_features(self, st, frame_idx):
        """Launch the next encoder batch on the side stream when the first frame without features is at most
        `async_lookahead` frames away in the propagation order."""
        order = st.get("_encode_order")
        if not order or frame_idx not in order or self.encode_batch <= 1:
            return
        cache, pend = st["cached_features"], st.setdefault("_pending_features", {})
        ahead = order[order.index(frame_idx) + 1:]
        have = set(st["images_idx"])
        missing = [t for t in ahead if t not in cache and t not in pend and t in have]
        if not missing or ahead.index(missing[0]) > self.async_lookahead:
            return
        n_batches = -(-len(missing) // self.encode_batch)
        todo = missing[: -(-len(missing) // n_batches)]
        if self._hip_enc is None:
            # a VIEW of the model: the same weights and weight planes (no copies), its own workspace arena
            self._hip_enc = HipSam2.view_of(self.hip)
            self._enc_stream = torch.cuda.Stream(device=self.device)
        if self._hip_enc.get_precision() != self.hip.get_precision():     # the mode is per model: keep the twin in step
            self._hip_enc.set_precision(self.hip.get_precision())
        main = torch.cuda.current_stream(self.device)
        es = self._enc_stream
        es.wait_stream(main)                       # the frames were ingested on the caller's stream
        # st["images"] was allocated on the caller's stream and is read here on the side stream: tell the caching
        # allocator, or add_new_frames / release_old_frames could free and re-use the block while the encoder reads it
        st["images"].record_stream(es)
        with torch.cuda.stream(es):
            if len(todo) == 1:
                feats = [self._hip_enc.image_encoder(st["images"][st["images_idx"].index(todo[0])])]
            else:
                pos = torch.tensor([st["images_idx"].index(t) for t in todo], device=st["images"].device)
                feats = self._hip_enc.image_encoder_batch(st["images"].index_select(0, pos))
            ev = torch.cuda.Event()
            ev.record(es)
        for t, ft in zip(todo, feats):
            pend[t] = (ft, ev)
        self.stats["encoder_runs"] += len(todo)
        self.stats["encoder_launches"] += 1

    def _drop_pending(self, st, t):
        """Forget a side-stream result that will not be adopted; its buffers return to the allocator only after the side
        stream is done with them (they were allocated under that stream)."""
        st["_pending_features"].pop(t, None)

    def encode_frames(self, st, frame_indices):
        """Encode the given retained frames now (batches of encode_batch) and return their pyramids in order; frames that
        are already cached are not encoded again.  Used by the pass-sharded driver, which encodes a buffer up front."""
        cache = st["cached_features"]
        todo = [t for t in frame_indices if t not in cache]
        for i in range(0, len(todo), max(self.encode_batch, 1)):
            chunk = todo[i:i + max(self.encode_batch, 1)]
            if len(chunk) == 1:
                feats = [self.hip.image_encoder(st["images"][st["images_idx"].index(chunk[0])])]
            else:
                pos = torch.tensor([st["images_idx"].index(t) for t in chunk], device=st["images"].device)
                feats = self.hip.image_encoder_batch(st["images"].index_select(0, pos))
            for t, ft in zip(chunk, feats):
                cache[t] = ft
            self.stats["encoder_runs"] += len(chunk)
            self.stats["encoder_launches"] += 1
        return [cache[t] for t in frame_indices]

    def feature_shapes(self):
        """Per-frame pyramid (fpn0, fpn1, fpn2), token-major fp32."""
        return [(65536, 32), (16384, 64), (4096, 256)]

    def entry_dims(self):
        """Shapes of a bank entry / level-2 feature (what a peer rank allocates before receiving one)."""
        return dict(tokens=4096, mem_dim=self.mem_dim, ptr_dim=self.hidden_dim, mask_side=256, feat_dim=self.hidden_dim)

    def _trim_feature_cache(self, st, keep=()):
        """The cache holds 16 MiB per frame; whatever the release policy is (max_inference_state_frames = -1 never
        releases), keep at most the frames of the current propagation window + one encode batch, evicting the oldest
        insertions first.  Pinned entries (preload-bank features without frames) are never evicted."""
        cache = st["cached_features"]
        cap = max(max(len(st.get("_encode_order") or ()), 1) + self.encode_batch + 1, st.get("_feature_cache_cap", 0))
        pinned = st.get("_pinned_features") or ()
        if len(cache) - len(pinned) <= cap:
            return
        for t in list(cache):
            if len(cache) - len(pinned) <= cap:
                break
            if t not in pinned and t not in keep:
                cache.pop(t)

    # ------------------------------------------------------------------ object table (A17)
    def _new_slot(self, st, obj_id):
        idx = len(st["obj_id_to_idx"])
        st["obj_id_to_idx"][obj_id] = idx
        st["obj_idx_to_id"][idx] = obj_id
        st["obj_ids"] = list(st["obj_id_to_idx"])
        st["point_inputs_per_obj"][idx] = {}
        st["mask_inputs_per_obj"][idx] = {}
        st["output_dict_per_obj"][idx] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        st["temp_output_dict_per_obj"][idx] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        return idx

    def _obj_id_to_idx(self, st, obj_id):
        """_obj_id_to_idx incl. the online new-object path (sam2_video_predictor.py:219-333)."""
        idx = st["obj_id_to_idx"].get(obj_id)
        if idx is not None:
            return idx
        idx = self._new_slot(st, obj_id)
        if st["tracking_has_started"]:
            od = st["output_dict"]
            inds = sorted(od["cond_frame_outputs"].keys())
            mx = st["max_update_length_for_new_obj_id"]
            if mx > 0:
                inds = inds[-mx:]
            for t in st["preloading_memory_cond_frame_idx"] or []:
                if t not in inds:
                    inds.append(t)
            for t in inds:
                cons = self._consolidate(st, t, True, True)
                od["cond_frame_outputs"][t] = cons
                self._add_output_per_object(st, t, cons, "cond_frame_outputs")
        return idx

    def _get_obj_num(self, st):
        return len(st["obj_idx_to_id"])

    # ------------------------------------------------------------------ prompts (A6)
    @torch.inference_mode()
    def add_new_points_or_box(self, inference_state, frame_idx, obj_id, points=None, labels=None, clear_old_points=True,
                              normalize_coords=True, box=None):
        """add_new_points_or_box (sam2_video_predictor.py:344-520)."""
        st = inference_state
        obj_idx = self._obj_id_to_idx(st, obj_id)
        if (points is not None) != (labels is not None):
            raise ValueError("points and labels must be provided together")
        if points is None and box is None:
            raise ValueError("at least one of points or box must be provided as input")
        points = torch.zeros(0, 2, dtype=torch.float32) if points is None else torch.as_tensor(points, dtype=torch.float32)
        labels = torch.zeros(0, dtype=torch.int32) if labels is None else torch.as_tensor(labels, dtype=torch.int32)
        if points.dim() == 2:
            points = points.unsqueeze(0)
        if labels.dim() == 1:
            labels = labels.unsqueeze(0)
        if box is not None:
            if not clear_old_points:
                raise ValueError("cannot add box without clearing old points, since box prompt must be provided "
                                 "before any point prompt (please use clear_old_points=True instead)")
            box = torch.as_tensor(box, dtype=torch.float32)
            points = torch.cat([box.reshape(1, 2, 2), points], dim=1)
            labels = torch.cat([torch.tensor([[2, 3]], dtype=torch.int32), labels], dim=1)
        if normalize_coords:
            points = points / torch.tensor([st["video_width"], st["video_height"]], dtype=torch.float32)
        points = points * self.image_size
        pin = {"point_coords": points.to(self.device), "point_labels": labels.to(self.device)}
        if not clear_old_points and frame_idx in st["point_inputs_per_obj"][obj_idx]:
            old = st["point_inputs_per_obj"][obj_idx][frame_idx]
            pin = {"point_coords": torch.cat([old["point_coords"], pin["point_coords"]], dim=1),
                   "point_labels": torch.cat([old["point_labels"], pin["point_labels"]], dim=1)}
        st["point_inputs_per_obj"][obj_idx][frame_idx] = pin
        st["mask_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        # a frame that was never tracked is an initial conditioning frame: the prompt alone produces the mask, no memory
        # (as in SAM).  Otherwise the prompt CORRECTS the tracked mask (sam2_video_predictor.py:428-433): the pass is
        # conditioned on this object's memory, in the direction the frame was tracked in, and its output replaces the
        # frame's non-conditioning entry (add_all_frames_to_correct_as_cond is false in every sam2.1 config).
        is_init_cond_frame = frame_idx not in st["frames_already_tracked"]
        reverse = False if is_init_cond_frame else st["frames_already_tracked"][frame_idx]["reverse"]
        storage_key = "cond_frame_outputs" if is_init_cond_frame else "non_cond_frame_outputs"
        obj_tmp, obj_out = st["temp_output_dict_per_obj"][obj_idx], st["output_dict_per_obj"][obj_idx]
        prev = obj_tmp[storage_key].get(frame_idx) or obj_out["cond_frame_outputs"].get(frame_idx) \
            or obj_out["non_cond_frame_outputs"].get(frame_idx)
        # a second prompt for the same object on the same frame (e.g. YOLO emits two boxes of one class) or a correction:
        # the previous prediction, clamped to [-32, 32], is fed back as a mask prompt (sam2_video_predictor.py:470-483)
        prev_logits = None
        if prev is not None and prev["pred_masks"] is not None:
            prev_logits = torch.clamp(prev["pred_masks"].to(self.device, torch.float32), -32.0, 32.0).reshape(1, 256, 256)
        f0, f1, f2 = self._prompt_features(st, frame_idx)
        npts = pin["point_labels"].shape[1]
        multimask = self.cfg.multimask_min_pt_num <= npts <= self.cfg.multimask_max_pt_num   # _use_multimask :922-932
        if is_init_cond_frame:
            # single-object SAM pass without memory: pix = feat + no_mem_embed (sam2_base.py:651-657)
            low, ptr, obj, _ = self.hip.sam_heads(1, f2, f0, f1, pin["point_coords"], pin["point_labels"], multimask,
                                                  pix_bcast=True, add_no_mem_embed=True, mask_inputs=prev_logits)
        else:
            # _run_single_frame_inference(output_dict=<this object's slice>, batch_size=1, is_init_cond_frame=False):
            # memory-conditioned features from the object's own bank entries, then the SAM heads with the prompt
            mem_entries, ptr_entries = self._bank_for_frame(st, frame_idx, 1, reverse, od=obj_out)
            pix = self._memory_conditioned(1, f2, mem_entries, ptr_entries)
            low, ptr, obj, _ = self.hip.sam_heads(1, pix, f0, f1, pin["point_coords"], pin["point_labels"], multimask,
                                                  mask_inputs=prev_logits)
        low = self._fill_holes(low)
        obj_tmp[storage_key][frame_idx] = {
            "maskmem_features": None, "maskmem_pos_enc": None, "pred_masks": low.unsqueeze(1),
            "obj_ptr": ptr, "object_score_logits": obj.unsqueeze(1)}
        cons = self._consolidate(st, frame_idx, is_init_cond_frame, False)
        return frame_idx, st["obj_ids"], self._video_res(st, cons["pred_masks"])

    def add_new_points(self, *a, **k):
        return self.add_new_points_or_box(*a, **k)

    @torch.inference_mode()
    def add_new_mask(self, inference_state, frame_idx, obj_id, mask):
        """add_new_mask (sam2_video_predictor.py:527-616): a user mask for one object on one frame.  The mask itself
        becomes the output (SAM2Base._use_mask_as_output): low-res logits = antialiased downsample of mask*20-10, the SAM
        heads (fed mask_downsample(mask) as mask prompt) only supply the object pointer, objectness = the mask is
        non-empty.  A mask of another size is resized to the model resolution (bilinear, antialias) and re-binarised at
        0.5 (:552-561)."""
        st = inference_state
        if not isinstance(mask, torch.Tensor):
            mask = torch.tensor(np.asarray(mask), dtype=torch.bool)
        assert mask.dim() == 2
        obj_idx = self._obj_id_to_idx(st, obj_id)
        S = self.image_size
        m = mask.to(self.device).to(torch.float32)[None].contiguous()
        if tuple(m.shape[-2:]) != (S, S):
            m = self.hip.resize_aa(m, S, S, threshold=0.5)
        # on an already-tracked frame the mask corrects the tracked output: same computation (the mask IS the output, no
        # memory is read: track_step :873-879), stored as the frame's non-conditioning entry (:583-586)
        is_init_cond_frame = frame_idx not in st["frames_already_tracked"]
        storage_key = "cond_frame_outputs" if is_init_cond_frame else "non_cond_frame_outputs"
        f0, f1, f2 = self._prompt_features(st, frame_idx)
        low, ptr, obj = self.hip.use_mask_as_output(1, f2, f0, f1, m)
        low = self._fill_holes(low)
        st["mask_inputs_per_obj"][obj_idx][frame_idx] = m[None]
        st["point_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        st["temp_output_dict_per_obj"][obj_idx][storage_key][frame_idx] = {
            "maskmem_features": None, "maskmem_pos_enc": None, "pred_masks": low.unsqueeze(1), "obj_ptr": ptr,
            "object_score_logits": obj.unsqueeze(1)}
        cons = self._consolidate(st, frame_idx, is_init_cond_frame, False)
        return frame_idx, st["obj_ids"], self._video_res(st, cons["pred_masks"])

    def _prompt_features(self, st, frame_idx):
        """Pyramid of a frame that receives a prompt.  A DS2BANK preload frame carries only its level-2 feature (enough
        for the A17 re-consolidation), not the image: it cannot be prompted."""
        f0, f1, f2 = self._get_image_feature(st, frame_idx)
        if f0 is None or f1 is None:
            raise RuntimeError(f"frame {frame_idx} is a preload-bank frame without its image: prompts on it are not possible")
        return f0, f1, f2

    def _video_res(self, st, low, packed=False):
        """_get_orig_video_res_output (sam2_video_predictor.py:618-642)."""
        logits, bits = self.hip.mask_output(low[:, 0].contiguous(), st["video_height"], st["video_width"],
                                            want_logits=not packed, want_packed=packed)
        return bits if packed else logits

    # ------------------------------------------------------------------ consolidation (A9)
    def _consolidate(self, st, frame_idx, is_cond, run_mem_encoder):
        """_consolidate_temp_output_across_obj (sam2_video_predictor.py:644-767) at low resolution."""
        B = self._get_obj_num(st)
        key = "cond_frame_outputs" if is_cond else "non_cond_frame_outputs"
        d = self.device
        cons = {"maskmem_features": None, "maskmem_pos_enc": None,
                "pred_masks": torch.full((B, 1, 256, 256), NO_OBJ_SCORE, dtype=torch.float32, device=d),
                "obj_ptr": torch.full((B, self.hidden_dim), NO_OBJ_SCORE, dtype=torch.float32, device=d),
                "object_score_logits": torch.full((B, 1), 10.0, dtype=torch.float32, device=d)}
        for i in range(B):
            tmp, od = st["temp_output_dict_per_obj"][i], st["output_dict_per_obj"][i]
            out = tmp[key].get(frame_idx) or od["cond_frame_outputs"].get(frame_idx) \
                or od["non_cond_frame_outputs"].get(frame_idx)
            if out is None:
                if run_mem_encoder:
                    # _get_empty_mask_ptr (:769-804): the reference runs the SAM heads on an all-zero mask and
                    # then multiplies the pointer by is_obj_appearing = 0 (sam2_base.py:436-444) => no_obj_ptr.
                    cons["obj_ptr"][i:i + 1] = self.hip.no_obj_ptr
                continue
            cons["pred_masks"][i:i + 1] = out["pred_masks"]
            cons["obj_ptr"][i:i + 1] = out["obj_ptr"]
            cons["object_score_logits"][i:i + 1] = out["object_score_logits"]
        if run_mem_encoder:
            _, _, f2 = self._get_image_feature(st, frame_idx)
            cons["maskmem_features"] = self.hip.memory_encoder(
                B, f2, cons["pred_masks"][:, 0].contiguous(), cons["object_score_logits"][:, 0].contiguous(),
                binarize=self.cfg.binarize_mask_from_pts_for_mem_enc)
        return cons

    def _add_output_per_object(self, st, frame_idx, out, key):
        """_add_output_per_object (sam2_video_predictor.py:1027-1058): per-object views."""
        for i, od in st["output_dict_per_obj"].items():
            s = slice(i, i + 1)
            if out["pred_masks"].shape[0] <= i:
                continue
            od[key][frame_idx] = {
                "maskmem_features": None if out["maskmem_features"] is None else out["maskmem_features"][s],
                "maskmem_pos_enc": None, "pred_masks": out["pred_masks"][s], "obj_ptr": out["obj_ptr"][s],
                "object_score_logits": out["object_score_logits"][s]}

    # ------------------------------------------------------------------ prompt / object removal
    def _frame_masks_after_edit(self, st, frame_idx):
        """video-resolution masks of every object on one frame after its prompts changed (no memory encoding)."""
        is_cond = any(frame_idx in tmp["cond_frame_outputs"] for tmp in st["temp_output_dict_per_obj"].values())
        cons = self._consolidate(st, frame_idx, is_cond, False)
        return self._video_res(st, cons["pred_masks"])

    @torch.inference_mode()
    def clear_all_prompts_in_frame(self, inference_state, frame_idx, obj_id, need_output=True):
        """clear_all_prompts_in_frame (sam2_video_predictor.py:1061-1131): forget the points / mask of one object on one
        frame.  A frame left without any object's input stops being a conditioning frame: its consolidated output is
        demoted to a non-conditioning one, and when no conditioning frame remains all tracking results are dropped."""
        st = inference_state
        obj_idx = self._obj_id_to_idx(st, obj_id)
        st["point_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        st["mask_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
            st["temp_output_dict_per_obj"][obj_idx][key].pop(frame_idx, None)
        has_input = any(frame_idx in st["point_inputs_per_obj"][i] or frame_idx in st["mask_inputs_per_obj"][i]
                        for i in range(self._get_obj_num(st)))
        if not has_input:
            od, cfi = st["output_dict"], st["consolidated_frame_inds"]
            cfi["cond_frame_outputs"].discard(frame_idx)
            cfi["non_cond_frame_outputs"].discard(frame_idx)
            out = od["cond_frame_outputs"].pop(frame_idx, None)
            if out is not None:
                od["non_cond_frame_outputs"][frame_idx] = out
                st["frames_already_tracked"].pop(frame_idx, None)
            for o in st["output_dict_per_obj"].values():
                obj_out = o["cond_frame_outputs"].pop(frame_idx, None)
                if obj_out is not None:
                    o["non_cond_frame_outputs"][frame_idx] = obj_out
            if len(od["cond_frame_outputs"]) == 0:
                self._reset_tracking_results(st)
        if not need_output:
            return None
        return frame_idx, st["obj_ids"], self._frame_masks_after_edit(st, frame_idx)

    @torch.inference_mode()
    def remove_object(self, inference_state, obj_id, strict=False, need_output=True):
        """remove_object (sam2_video_predictor.py:1438-1549): take one object out of the tracking state.  Its prompts are
        cleared first (which may demote conditioning frames), the object table is renumbered, and its row is cut out of
        every stored batch entry (memory features, low-res masks, pointers, objectness).  Returns (obj_ids,
        [(frame_idx, video_res_masks)] for the frames the object had prompts on)."""
        st = inference_state
        rm = st["obj_id_to_idx"].get(obj_id)
        updated = []
        if rm is None:
            if strict:
                raise RuntimeError(f"Cannot remove object id {obj_id} as it doesn't exist. "
                                   f"All existing object ids: {st['obj_ids']}.")
            return st["obj_ids"], updated
        if len(st["obj_id_to_idx"]) == 1:
            self.reset_state(st)
            return st["obj_ids"], updated
        input_frames = set(st["point_inputs_per_obj"][rm]) | set(st["mask_inputs_per_obj"][rm])
        for t in input_frames:
            self.clear_all_prompts_in_frame(st, t, obj_id, need_output=False)
        n_old = len(st["obj_ids"])
        keep = [i for i in range(n_old) if i != rm]
        ids = [st["obj_ids"][i] for i in keep]
        st["obj_id_to_idx"] = {o: i for i, o in enumerate(ids)}
        st["obj_idx_to_id"] = dict(enumerate(ids))
        st["obj_ids"] = ids
        for name in ("point_inputs_per_obj", "mask_inputs_per_obj", "output_dict_per_obj", "temp_output_dict_per_obj"):
            vals = [st[name].pop(i) for i in range(n_old)]
            st[name].update((new, vals[old]) for new, old in enumerate(keep))
        sel = torch.tensor(keep, dtype=torch.long)
        for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
            for t, out in st["output_dict"][key].items():
                for f in ("maskmem_features", "pred_masks", "obj_ptr", "object_score_logits"):
                    if out[f] is not None:
                        out[f] = out[f].index_select(0, sel.to(out[f].device))
                self._add_output_per_object(st, t, out, key)
        if need_output:
            updated = [(t, self._frame_masks_after_edit(st, t)) for t in input_frames]
        return st["obj_ids"], updated

    # ------------------------------------------------------------------ propagate (A9, A10, A11)
    @torch.inference_mode()
    def propagate_in_video_preflight(self, inference_state):
        """propagate_in_video_preflight (sam2_video_predictor.py:807-893)."""
        st = inference_state
        st["tracking_has_started"] = True
        od, cfi = st["output_dict"], st["consolidated_frame_inds"]
        for is_cond in (False, True):
            key = "cond_frame_outputs" if is_cond else "non_cond_frame_outputs"
            inds = set()
            for tmp in st["temp_output_dict_per_obj"].values():
                inds.update(tmp[key].keys())
            cfi[key].update(inds)
            for t in inds:
                cons = self._consolidate(st, t, is_cond, True)
                od[key][t] = cons
                self._add_output_per_object(st, t, cons, key)
            for tmp in st["temp_output_dict_per_obj"].values():
                tmp[key].clear()
        for t in od["cond_frame_outputs"]:
            od["non_cond_frame_outputs"].pop(t, None)
        for o in st["output_dict_per_obj"].values():
            for t in o["cond_frame_outputs"]:
                o["non_cond_frame_outputs"].pop(t, None)
        for t in cfi["cond_frame_outputs"]:
            assert t in od["cond_frame_outputs"]
            cfi["non_cond_frame_outputs"].discard(t)

    def _select_cond(self, frame_idx, cond, preload_idx):
        """select_closest_cond_frames (sam2_utils.py:19-66), Det-SAM2 variant."""
        mx = self.cfg.max_cond_frames_in_attn
        if mx == -1 or len(cond) <= mx:
            return cond, {}
        sel = {}
        before = max((t for t in cond if t < frame_idx), default=None)
        if before is not None:
            sel[before] = cond[before]
        after = min((t for t in cond if t >= frame_idx), default=None)
        if after is not None:
            sel[after] = cond[after]
        rest = sorted((t for t in cond if t not in sel), key=lambda x: abs(x - frame_idx))[: mx - len(sel)]
        sel.update((t, cond[t]) for t in rest)
        for t in preload_idx or []:
            if t not in sel:
                sel[t] = cond[t]
        return sel, {t: v for t, v in cond.items() if t not in sel}

    def _bank_for_frame(self, st, frame_idx, B, reverse, od=None):
        """Index logic of _prepare_memory_conditioned_features (sam2_base.py:500-648): which stored entries
        enter the bank, their temporal slots and the pointer list.  Returns (mem_entries, ptr_entries).
        ``od``: the output dict to read - the batch dict (tracking) or one object's slice (correction prompts, B = 1)."""
        od = st["output_dict"] if od is None else od
        if len(od["cond_frame_outputs"]) == 0:
            raise RuntimeError(f"frame {frame_idx}: no conditioning frame to attend to (sam2_base.py:516)")
        sign = -1 if reverse else 1
        sel, unsel = self._select_cond(frame_idx, od["cond_frame_outputs"], st["preloading_memory_cond_frame_idx"])
        slots = [(0, t, out) for t, out in sel.items()]
        for t_pos in range(1, self.num_maskmem):
            t_rel = self.num_maskmem - t_pos
            prev = frame_idx + t_rel if reverse else frame_idx - t_rel
            out = od["non_cond_frame_outputs"].get(prev)
            if out is None:
                out = unsel.get(prev)
            slots.append((t_pos, prev, out))
        mem_entries, tr = [], {"frame": frame_idx, "mem": [], "ptr": []}
        for t_pos, t, out in slots:
            if out is None:
                continue
            f = out["maskmem_features"]
            if f.shape[0] != B:
                raise RuntimeError(f"memory entry of frame {t} has batch {f.shape[0]}, expected {B}")
            mem_entries.append((f, self.num_maskmem - t_pos - 1))
            tr["mem"].append((t_pos, t))
        max_ptrs = min(st["num_frames"], self.cfg.max_obj_ptrs_in_encoder)
        ptr_cond = {t: o for t, o in sel.items() if (t >= frame_idx if reverse else t <= frame_idx)}
        pos_ptrs = [((frame_idx - t) * sign, o["obj_ptr"]) for t, o in ptr_cond.items()]
        for t_diff in range(1, max_ptrs):
            t = frame_idx + t_diff if reverse else frame_idx - t_diff
            if t < 0 or t >= st["num_frames"]:
                break
            o = od["non_cond_frame_outputs"].get(t, unsel.get(t))
            if o is not None:
                pos_ptrs.append((t_diff, o["obj_ptr"]))
        tdm = np.float32(max_ptrs - 1)
        ptr_entries = [(p, float(np.float32(pos) / tdm)) for pos, p in pos_ptrs]
        tr["ptr"] = [p for p, _ in pos_ptrs]
        tr["nk"], tr["n_ptr_tok"] = 4096 * len(mem_entries) + 4 * len(ptr_entries), 4 * len(ptr_entries)
        if self.trace is not None:
            self.trace.append(tr)
        return mem_entries, ptr_entries

    def _memory_conditioned(self, B, f2, mem_entries, ptr_entries):
        """_prepare_memory_conditioned_features' tensor part (sam2_base.py:565-690): bank -> memory attention.  One fused call when the
        stage interface has it (the fp32 memory / memory_pos tensors are then never materialised), else the two stages."""
        if not mem_entries:
            raise RuntimeError("memory attention needs at least one memory frame in the bank (a tracked frame always attends its "
                               "conditioning frame, sam2_base.py:565-590); got object pointers only")
        fused = getattr(self.hip, "bank_attention", None)
        if fused is not None:
            return fused(B, f2, mem_entries, ptr_entries)
        memory, memory_pos = self.hip.bank_assemble(B, mem_entries, ptr_entries)
        return self.hip.memory_attention(B, f2, memory, memory_pos, 4 * len(ptr_entries))

    def _fill_holes(self, low):
        """fill_holes_in_mask_scores on the low-res logits [B,256,256] (sam2_video_predictor.py:1343-1346)."""
        if self.fill_hole_area <= 0:
            return low
        from .misc import fill_holes_in_mask_scores
        return fill_holes_in_mask_scores(low, self.fill_hole_area)

    def _track_frame(self, st, frame_idx, B, reverse):
        """_run_single_frame_inference + track_step for a non-conditioning frame (is_init_cond_frame=False,
        no prompts, run_mem_encoder=True)  (sam2_video_predictor.py:1280-1365; sam2_base.py:857-919)."""
        f0, f1, f2 = self._get_image_feature(st, frame_idx)
        mem_entries, ptr_entries = self._bank_for_frame(st, frame_idx, B, reverse)
        pix = self._memory_conditioned(B, f2, mem_entries, ptr_entries)
        low, ptr, obj, _ = self.hip.sam_heads(B, pix, f0, f1, None, None, multimask=True)   # num_pts=0 => multimask
        mem = self.hip.memory_encoder(B, f2, low, obj, binarize=False)
        low = self._fill_holes(low)   # after the memory encoder, as in _run_single_frame_inference (:1343-1346)
        self.stats["tracked_frames"] += 1
        return {"maskmem_features": mem, "maskmem_pos_enc": None, "pred_masks": low.unsqueeze(1), "obj_ptr": ptr,
                "object_score_logits": obj.unsqueeze(1)}

    @torch.inference_mode()
    def propagate_in_video(self, inference_state, start_frame_idx=None, max_frame_num_to_track=None, reverse=False,
                           output="logits"):
        """propagate_in_video (sam2_video_predictor.py:911-1025).  Generator of
        (frame_idx, obj_ids, video_res_masks fp32 [B,1,Hv,Wv]); with output='packed' the third item is the
        thresholded mask packed 8 px/byte [B,Hv,Wv/8] (the det_sam2_RT.py:396-399 consumer only needs bits)."""
        st = inference_state
        self.propagate_in_video_preflight(st)
        od, cfi = st["output_dict"], st["consolidated_frame_inds"]
        obj_ids, n, B = st["obj_ids"], st["num_frames"], self._get_obj_num(st)
        if len(od["cond_frame_outputs"]) == 0:
            raise RuntimeError("No points are provided; please add points first")
        if start_frame_idx is None:
            start_frame_idx = min(od["cond_frame_outputs"])
        if max_frame_num_to_track is None:
            max_frame_num_to_track = n
        if reverse:
            end = max(start_frame_idx - max_frame_num_to_track + 1, 0)
            order = range(start_frame_idx, end - 1, -1) if start_frame_idx > 0 else []
        else:
            end = min(start_frame_idx + max_frame_num_to_track, n - 1)
            order = range(start_frame_idx, end + 1)
        st["_encode_order"] = list(order)   # lets the feature cache batch-encode upcoming frames
        try:
            yield from self._propagate_loop(st, order, od, cfi, obj_ids, B, reverse, output)
        finally:                            # generator exhausted, closed or abandoned: nothing stays queued on the side
            for t in list(st.get("_pending_features") or {}):
                self._drop_pending(st, t)

    def _propagate_loop(self, st, order, od, cfi, obj_ids, B, reverse, output):
        for t in order:
            if t in cfi["cond_frame_outputs"]:
                key = "cond_frame_outputs"
                cur = od[key][t]
            elif t in cfi["non_cond_frame_outputs"]:
                key = "non_cond_frame_outputs"
                cur = od[key][t]
            else:
                key = "non_cond_frame_outputs"
                cur = self._track_frame(st, t, B, reverse)
                od[key][t] = cur
            self._add_output_per_object(st, t, cur, key)
            st["frames_already_tracked"][t] = {"reverse": reverse}
            yield t, obj_ids, self._video_res(st, cur["pred_masks"], packed=(output == "packed"))

    # ------------------------------------------------------------------ eviction / reset (A16)
    def release_old_frames(self, inference_state, frame_idx, max_inference_state_frames, pre_frames, release_images=False):
        """release_old_frames (sam2_video_predictor.py:1215-1273); also drops the cached features of released frames."""
        st = inference_state
        oldest = frame_idx - max_inference_state_frames
        od = st["output_dict"]
        old_c = [t for t in od["cond_frame_outputs"] if pre_frames - 1 < t <= oldest]
        old_n = [t for t in od["non_cond_frame_outputs"] if pre_frames - 1 < t <= oldest]
        for t in old_n:
            od["non_cond_frame_outputs"].pop(t, None)
            for o in st["output_dict_per_obj"].values():
                o["non_cond_frame_outputs"].pop(t, None)
        for t in old_c:
            od["cond_frame_outputs"].pop(t, None)
            st["consolidated_frame_inds"]["cond_frame_outputs"].discard(t)
            for o in st["output_dict_per_obj"].values():
                o["cond_frame_outputs"].pop(t, None)
        for t in [t for t in st["cached_features"] if pre_frames - 1 < t <= oldest]:
            st["cached_features"].pop(t, None)
            if st.get("_pinned_features"):
                st["_pinned_features"].discard(t)
        pend = st.get("_pending_features") or {}
        for t in [t for t in pend if pre_frames - 1 < t <= oldest]:     # encoded ahead but never adopted
            self._drop_pending(st, t)
        if release_images:
            old = [t for t in st["images_idx"] if pre_frames - 1 < t <= oldest]
            rm = {st["images_idx"].index(t) for t in old}
            keep = torch.tensor([i for i in range(st["images"].size(0)) if i not in rm], dtype=torch.long, device=st["images"].device)
            st["images"] = torch.index_select(st["images"], 0, keep)
            st["images_idx"] = [t for t in st["images_idx"] if t not in old]
            assert len(st["images"]) == len(st["images_idx"])

    @torch.inference_mode()
    def reset_state(self, inference_state):
        """reset_state (sam2_video_predictor.py:1134-1145): tracking results AND the object table."""
        st = inference_state
        self._reset_tracking_results(st)
        for k in ("point_inputs_per_obj", "mask_inputs_per_obj", "output_dict_per_obj", "temp_output_dict_per_obj"):
            st[k].clear()
        st["obj_id_to_idx"].clear()
        st["obj_idx_to_id"].clear()
        st["obj_ids"].clear()

    def _reset_tracking_results(self, st):
        """_reset_tracking_results (sam2_video_predictor.py:1147-1172): every prompt and output, the object table stays."""
        for name in ("point_inputs_per_obj", "mask_inputs_per_obj"):
            for v in st[name].values():
                v.clear()
        for name in ("output_dict_per_obj", "temp_output_dict_per_obj"):
            for v in st[name].values():
                v["cond_frame_outputs"].clear()
                v["non_cond_frame_outputs"].clear()
        st["output_dict"]["cond_frame_outputs"].clear()
        st["output_dict"]["non_cond_frame_outputs"].clear()
        st["consolidated_frame_inds"]["cond_frame_outputs"].clear()
        st["consolidated_frame_inds"]["non_cond_frame_outputs"].clear()
        st["tracking_has_started"] = False
        st["frames_already_tracked"].clear()
