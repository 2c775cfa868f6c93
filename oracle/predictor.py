"""ORACLE (test infrastructure, not product): fp32 PyTorch-CPU restatement of the tracking
step (``SAM2Base``), the predictor state machine (``SAM2VideoPredictor``) and frame ingest
-- SURVEY.md section 8a rows A3, A6, A9, A10, A11, A15, A16, A17, A18.

Numerics embodied: pure fp32, fp16 frame storage with in-place fp16
normalisation (misc.py:328,358-359), bf16 round-trip of ``maskmem_features``
(sam2_video_predictor.py:1337,1396; sam2_base.py:577), and NO hole filling (the reference's
``fill_holes_in_mask_scores`` silently does nothing without its CUDA extension, misc.py:389-391).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg import this.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import modeling as M
from .modeling import NO_OBJ_SCORE


def load_frames(frames, image_size=1024, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """load_video_frames, list-of-ndarray branch (misc.py:280-284, 328-342, 358-359).  cv2.resize is restated in
    oracle/resize.py (parity-unpinned: cv2 is not installed here); identity for image_size x image_size frames."""
    from .resize import cv2_resize_linear_u8
    if isinstance(frames, np.ndarray):
        frames = [frames]
    images = torch.zeros(len(frames), 3, image_size, image_size, dtype=torch.float16)
    for n, fr in enumerate(frames):
        images[n] = torch.from_numpy(cv2_resize_linear_u8(fr, (image_size, image_size)) / 255.0).permute(2, 0, 1)
    images -= torch.tensor(mean, dtype=torch.float32)[:, None, None]
    images /= torch.tensor(std, dtype=torch.float32)[:, None, None]
    return images, frames[0].shape[0], frames[0].shape[1]


def select_closest_cond_frames(frame_idx, cond, max_num, preload_idx=None):
    """sam2_utils.py:19-66 (Det-SAM2 variant that force-includes preload cond frames)."""
    if max_num == -1 or len(cond) <= max_num:
        return cond, {}
    sel = {}
    before = max((t for t in cond if t < frame_idx), default=None)
    if before is not None:
        sel[before] = cond[before]
    after = min((t for t in cond if t >= frame_idx), default=None)
    if after is not None:
        sel[after] = cond[after]
    remain = sorted((t for t in cond if t not in sel), key=lambda x: abs(x - frame_idx))[: max_num - len(sel)]
    sel.update((t, cond[t]) for t in remain)
    if preload_idx is not None:
        for t in preload_idx:
            if t not in sel:
                sel[t] = cond[t]
    return sel, {t: v for t, v in cond.items() if t not in sel}


class OraclePredictor:
    """Functional counterpart of SAM2VideoPredictor(SAM2Base) for the hot path."""

    def __init__(self, sd, cfg, fill_hole_area=0):
        self.sd, self.cfg = sd, cfg
        # 0 = what the CPU reference does (its CUDA-only extension is missing => misc.py:389-391 skips the step);
        # > 0 = the reference's behaviour WITH the extension, restated in oracle/cc.py
        self.fill_hole_area = fill_hole_area
        self.image_size = cfg.image_size
        self.hidden_dim, self.mem_dim = cfg.d_model, cfg.mem_dim
        self.num_maskmem = cfg.num_maskmem
        self.trace = None  # optional list collecting bank-selection traces (tests)

    # ------------------------------------------------------------------ SAM heads (A7+A8 glue)
    def forward_sam_heads(self, feats, point_inputs=None, mask_inputs=None, high_res=None, multimask=False):
        """_forward_sam_heads (sam2_base.py:254-397)."""
        sd, cfg = self.sd, self.cfg
        B = feats.size(0)
        if point_inputs is not None:
            coords, labels = point_inputs["point_coords"], point_inputs["point_labels"]
        else:
            coords = torch.zeros(B, 1, 2)
            labels = -torch.ones(B, 1, dtype=torch.int32)
        mask_prompt = None
        if mask_inputs is not None:
            msz = (4 * cfg.feat_hw, 4 * cfg.feat_hw)
            mask_prompt = mask_inputs if mask_inputs.shape[-2:] == msz else F.interpolate(
                mask_inputs.float(), size=msz, align_corners=False, mode="bilinear", antialias=True)
        sparse, dense = M.prompt_encoder(sd, cfg, coords, labels, mask_prompt)
        low, ious, toks, obj_logits = M.mask_decoder(sd, cfg, feats, M.dense_pe(sd, cfg), sparse, dense, multimask, high_res)
        appearing = obj_logits > 0
        low = torch.where(appearing[:, None, None], low, NO_OBJ_SCORE).float()
        high = F.interpolate(low, size=(cfg.image_size, cfg.image_size), mode="bilinear", align_corners=False)
        tok = toks[:, 0]
        if multimask:
            best = torch.argmax(ious, dim=-1)
            bi = torch.arange(B)
            low_best, high_best = low[bi, best].unsqueeze(1), high[bi, best].unsqueeze(1)
            if toks.size(1) > 1:
                tok = toks[bi, best]
        else:
            low_best, high_best = low, high
        ptr = M.mlp(sd, "obj_ptr_proj", tok, 3)
        lam = appearing.float()
        ptr = lam * ptr + (1 - lam) * sd["no_obj_ptr"]
        return low, high, ious, low_best, high_best, ptr, obj_logits

    def use_mask_as_output(self, feats, high_res, mask_inputs):
        """_use_mask_as_output (sam2_base.py:399-448)."""
        sd = self.sd
        mf = mask_inputs.float()
        high = mf * 20.0 - 10.0
        low = F.interpolate(high, size=(high.size(-2) // 4, high.size(-1) // 4), align_corners=False,
                            mode="bilinear", antialias=True)
        ious = mask_inputs.new_ones(mask_inputs.size(0), 1).float()
        ds = F.conv2d(mf, sd["mask_downsample.weight"], sd["mask_downsample.bias"], stride=4)
        ptr = self.forward_sam_heads(feats, mask_inputs=ds, high_res=high_res)[5]
        lam = torch.any(mask_inputs.flatten(1).float() > 0.0, dim=1)[..., None].float()
        obj_logits = 20.0 * lam - 10.0
        ptr = lam * ptr + (1 - lam) * sd["no_obj_ptr"]
        return low, high, ious, low, high, ptr, obj_logits

    # ------------------------------------------------------------------ memory bank (A11)
    def memory_conditioned_features(self, frame_idx, is_init_cond, feat, feat_pos, output_dict, num_frames,
                                    reverse, preload_idx):
        """_prepare_memory_conditioned_features (sam2_base.py:479-690). feat/feat_pos [HW,B,C]."""
        sd, cfg = self.sd, self.cfg
        B, C, H = feat.size(1), self.hidden_dim, cfg.feat_hw
        if is_init_cond:  # directly_add_no_mem_embed (sam2_base.py:651-657)
            return (feat + sd["no_mem_embed"]).permute(1, 2, 0).view(B, C, H, H)
        sign = -1 if reverse else 1
        mems, poss = [], []
        sel, unsel = select_closest_cond_frames(frame_idx, output_dict["cond_frame_outputs"],
                                                cfg.max_cond_frames_in_attn, preload_idx)
        slots = [(0, t, out) for t, out in sel.items()]
        for t_pos in range(1, self.num_maskmem):
            t_rel = self.num_maskmem - t_pos
            prev = frame_idx + t_rel if reverse else frame_idx - t_rel   # stride 1 (sam2_base.py:536-554)
            out = output_dict["non_cond_frame_outputs"].get(prev, None)
            if out is None:
                out = unsel.get(prev, None)
            slots.append((t_pos, prev, out))
        tr = {"frame": frame_idx, "mem": [], "ptr": []}
        for t_pos, t, prev in slots:
            if prev is None:
                continue
            tr["mem"].append((t_pos, t))
            f = prev["maskmem_features"].to(torch.float32)
            mems.append(f.flatten(2).permute(2, 0, 1))
            pe = prev["maskmem_pos_enc"][-1].flatten(2).permute(2, 0, 1)
            poss.append(pe + sd["maskmem_tpos_enc"][self.num_maskmem - t_pos - 1])
        max_ptrs = min(num_frames, cfg.max_obj_ptrs_in_encoder)
        ptr_cond = {t: o for t, o in sel.items() if (t >= frame_idx if reverse else t <= frame_idx)}
        pos_ptrs = [((frame_idx - t) * sign, o["obj_ptr"]) for t, o in ptr_cond.items()]
        for t_diff in range(1, max_ptrs):
            t = frame_idx + t_diff if reverse else frame_idx - t_diff
            if t < 0 or (num_frames is not None and t >= num_frames):
                break
            o = output_dict["non_cond_frame_outputs"].get(t, unsel.get(t, None))
            if o is not None:
                pos_ptrs.append((t_diff, o["obj_ptr"]))
        n_ptr_tok = 0
        if pos_ptrs:
            pos_list, ptr_list = zip(*pos_ptrs)
            tr["ptr"] = list(pos_list)
            ptrs = torch.stack(ptr_list, dim=0)
            op = M.sine_pe_1d(torch.tensor(pos_list) / (max_ptrs - 1), C)
            op = M.linear(sd, "obj_ptr_tpos_proj", op).unsqueeze(1).expand(-1, B, self.mem_dim)
            ptrs = ptrs.reshape(-1, B, C // self.mem_dim, self.mem_dim).permute(0, 2, 1, 3).flatten(0, 1)
            op = op.repeat_interleave(C // self.mem_dim, dim=0)
            mems.append(ptrs)
            poss.append(op)
            n_ptr_tok = ptrs.shape[0]
        memory, memory_pos = torch.cat(mems, 0), torch.cat(poss, 0)
        tr["nk"], tr["n_ptr_tok"] = memory.shape[0], n_ptr_tok
        if self.trace is not None:
            self.trace.append(tr)
        out = M.memory_attention(sd, cfg, feat, feat_pos, memory, memory_pos, n_ptr_tok)
        return out.permute(1, 2, 0).view(B, C, H, H)

    def encode_new_memory(self, feat, high_res_masks, obj_logits, is_mask_from_pts):
        """_encode_new_memory (sam2_base.py:692-743)."""
        sd, cfg = self.sd, self.cfg
        B, H = feat.size(1), cfg.feat_hw
        pix = feat.permute(1, 2, 0).view(B, self.hidden_dim, H, H)
        if cfg.binarize_mask_from_pts_for_mem_enc and is_mask_from_pts:
            m = (high_res_masks > 0).float()
        else:
            m = torch.sigmoid(high_res_masks)
        m = m * cfg.sigmoid_scale_for_mem_enc + cfg.sigmoid_bias_for_mem_enc
        f, pos = M.memory_encoder(sd, cfg, pix, m)
        lam = (obj_logits > 0).float()
        f = f + (1 - lam[..., None, None]) * sd["no_obj_embed_spatial"][..., None, None].expand(*f.shape)
        return f, [pos]

    # ------------------------------------------------------------------ track_step (A10)
    def track_step(self, frame_idx, is_init_cond, feats, poss, point_inputs, mask_inputs, output_dict, num_frames,
                   reverse=False, run_mem_encoder=True, prev_sam_mask_logits=None, preload_idx=None):
        """track_step/_track_step/_encode_memory_in_output (sam2_base.py:857-919, 746-840).
        feats/poss: lists of 3 [HW,B,C] tensors (levels 0,1,2)."""
        cfg = self.cfg
        sizes = [(4 * cfg.feat_hw, 4 * cfg.feat_hw), (2 * cfg.feat_hw, 2 * cfg.feat_hw)]
        high_res = [x.permute(1, 2, 0).view(x.size(1), x.size(2), *s) for x, s in zip(feats[:-1], sizes)]
        out = {"point_inputs": point_inputs, "mask_inputs": mask_inputs}
        if mask_inputs is not None:
            pix = feats[-1].permute(1, 2, 0).view(-1, self.hidden_dim, cfg.feat_hw, cfg.feat_hw)
            sam = self.use_mask_as_output(pix, high_res, mask_inputs)
        else:
            pix = self.memory_conditioned_features(frame_idx, is_init_cond, feats[-1], poss[-1], output_dict,
                                                   num_frames, reverse, preload_idx)
            if prev_sam_mask_logits is not None:
                mask_inputs = prev_sam_mask_logits
            npts = 0 if point_inputs is None else point_inputs["point_labels"].size(1)
            multimask = cfg.multimask_min_pt_num <= npts <= cfg.multimask_max_pt_num  # _use_multimask :922-932
            sam = self.forward_sam_heads(pix, point_inputs, mask_inputs, high_res, multimask)
        _, _, _, low, high, ptr, obj_logits = sam
        out.update(pred_masks=low, pred_masks_high_res=high, obj_ptr=ptr, object_score_logits=obj_logits)
        if run_mem_encoder:
            out["maskmem_features"], out["maskmem_pos_enc"] = self.encode_new_memory(
                feats[-1], high, obj_logits, point_inputs is not None)
        else:
            out["maskmem_features"] = out["maskmem_pos_enc"] = None
        return out

    # ------------------------------------------------------------------ state (A3, A16)
    def init_state(self, frames):
        """init_state (sam2_video_predictor.py:44-120)."""
        images, vh, vw = load_frames(frames, self.image_size)
        st = dict(images=images, num_frames=len(images), images_idx=list(range(len(images))),
                  video_height=vh, video_width=vw, point_inputs_per_obj={}, mask_inputs_per_obj={},
                  cached_features={}, constants={}, obj_id_to_idx=OrderedDict(), obj_idx_to_id=OrderedDict(),
                  obj_ids=[], output_dict={"cond_frame_outputs": {}, "non_cond_frame_outputs": {}},
                  output_dict_per_obj={}, temp_output_dict_per_obj={},
                  consolidated_frame_inds={"cond_frame_outputs": set(), "non_cond_frame_outputs": set()},
                  tracking_has_started=False, frames_already_tracked={},
                  preloading_memory_cond_frame_idx=None, preloading_memory_non_cond_frames_idx=None,
                  max_update_length_for_new_obj_id=100)
        self.image_feature(st, 0, 1)
        return st

    def update_state(self, frames, st):
        """update_state (sam2_video_predictor.py:160-204)."""
        new, vh, vw = load_frames(frames, self.image_size)
        assert (vh, vw) == (st["video_height"], st["video_width"])
        last = st["images_idx"][-1]
        st["images_idx"].extend(range(last + 1, last + 1 + len(new)))
        st["images"] = torch.cat((st["images"], new), dim=0)
        st["num_frames"] += len(new)
        return st

    def image_feature(self, st, frame_idx, batch):
        """_get_image_feature (sam2_video_predictor.py:1174-1212) + _prepare_backbone_features
        (sam2_base.py:463-477): returns (feats[3], pos[3]) each [HW,B,C]."""
        cached = st["cached_features"].get(frame_idx)
        if cached is None:
            img = st["images"][st["images_idx"].index(frame_idx)].float().unsqueeze(0)
            cached = M.forward_image(self.sd, self.cfg, img)
            st["cached_features"] = {frame_idx: cached}
        fpn, pos = cached
        feats = [f.expand(batch, -1, -1, -1).flatten(2).permute(2, 0, 1) for f in fpn]
        poss = [p.expand(batch, -1, -1, -1).flatten(2).permute(2, 0, 1) for p in pos]
        return feats, poss

    def _new_obj_slot(self, st, obj_id):
        idx = len(st["obj_id_to_idx"])
        st["obj_id_to_idx"][obj_id] = idx
        st["obj_idx_to_id"][idx] = obj_id
        st["obj_ids"] = list(st["obj_id_to_idx"])
        st["point_inputs_per_obj"][idx] = {}
        st["mask_inputs_per_obj"][idx] = {}
        st["output_dict_per_obj"][idx] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        st["temp_output_dict_per_obj"][idx] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        return idx

    def obj_id_to_idx(self, st, obj_id):
        """_obj_id_to_idx incl. Det-SAM2's online new-object path (sam2_video_predictor.py:219-333)."""
        idx = st["obj_id_to_idx"].get(obj_id)
        if idx is not None:
            return idx
        idx = self._new_obj_slot(st, obj_id)
        if st["tracking_has_started"]:  # A17: re-consolidate latest cond frames (+ preload) at the larger batch
            od = st["output_dict"]
            inds = sorted(od["cond_frame_outputs"].keys())
            mx = st["max_update_length_for_new_obj_id"]
            if mx > 0:
                inds = inds[-mx:]
            for t in st["preloading_memory_cond_frame_idx"] or []:
                if t not in inds:
                    inds.append(t)
            for t in inds:
                cons = self.consolidate(st, t, True, True, False)
                od["cond_frame_outputs"][t] = cons
                self.add_output_per_object(st, t, cons, "cond_frame_outputs")
        return idx

    # ------------------------------------------------------------------ prompts (A6)
    def add_new_points_or_box(self, st, frame_idx, obj_id, points=None, labels=None, box=None):
        """add_new_points_or_box, clear_old_points=True, normalize_coords=True
        (sam2_video_predictor.py:344-520)."""
        obj_idx = self.obj_id_to_idx(st, obj_id)
        if (points is not None) != (labels is not None):
            raise ValueError("points and labels must be provided together")
        if points is None and box is None:
            raise ValueError("at least one of points or box must be provided")
        points = torch.zeros(0, 2) if points is None else torch.as_tensor(points, dtype=torch.float32)
        labels = torch.zeros(0, dtype=torch.int32) if labels is None else torch.as_tensor(labels, dtype=torch.int32)
        if points.dim() == 2:
            points = points.unsqueeze(0)
        if labels.dim() == 1:
            labels = labels.unsqueeze(0)
        if box is not None:
            box = torch.as_tensor(box, dtype=torch.float32)
            points = torch.cat([box.reshape(1, 2, 2), points], dim=1)
            labels = torch.cat([torch.tensor([[2, 3]], dtype=torch.int32), labels], dim=1)
        points = points / torch.tensor([st["video_width"], st["video_height"]])
        points = points * self.image_size
        pin = {"point_coords": points, "point_labels": labels}
        st["point_inputs_per_obj"][obj_idx][frame_idx] = pin
        st["mask_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        # never tracked => initial conditioning frame (no memory); else a correction conditioned on this object's memory,
        # stored as a non-conditioning output (:428-433; add_all_frames_to_correct_as_cond is false)
        is_init = frame_idx not in st["frames_already_tracked"]
        reverse = False if is_init else st["frames_already_tracked"][frame_idx]["reverse"]
        key = "cond_frame_outputs" if is_init else "non_cond_frame_outputs"
        obj_out, obj_tmp = st["output_dict_per_obj"][obj_idx], st["temp_output_dict_per_obj"][obj_idx]
        prev = obj_tmp[key].get(frame_idx) or obj_out["cond_frame_outputs"].get(frame_idx) \
            or obj_out["non_cond_frame_outputs"].get(frame_idx)
        prev_logits = None
        if prev is not None and prev["pred_masks"] is not None:
            prev_logits = torch.clamp(prev["pred_masks"], -32.0, 32.0)
        cur, _ = self.single_frame(st, obj_out, frame_idx, 1, is_init, pin, None, reverse, False, prev_logits)
        obj_tmp[key][frame_idx] = cur
        cons = self.consolidate(st, frame_idx, is_init, False, True)
        return frame_idx, st["obj_ids"], self.video_res(st, cons["pred_masks_video_res"])

    def add_new_mask(self, st, frame_idx, obj_id, mask):
        """add_new_mask (sam2_video_predictor.py:527-616) (on a tracked frame the output is stored as a non-conditioning entry): the mask (resized with antialiasing
        to the model resolution and re-binarised at 0.5 when its size differs, :552-561) IS the output
        (_use_mask_as_output, sam2_base.py:399-448); the SAM heads only supply the object pointer."""
        obj_idx = self.obj_id_to_idx(st, obj_id)
        mask = torch.as_tensor(mask, dtype=torch.bool)
        assert mask.dim() == 2
        m = mask[None, None].float()
        if m.shape[-2:] != (self.image_size, self.image_size):
            m = F.interpolate(m, size=(self.image_size, self.image_size), align_corners=False, mode="bilinear", antialias=True)
            m = (m >= 0.5).float()
        st["mask_inputs_per_obj"][obj_idx][frame_idx] = m
        st["point_inputs_per_obj"][obj_idx].pop(frame_idx, None)
        is_init = frame_idx not in st["frames_already_tracked"]
        reverse = False if is_init else st["frames_already_tracked"][frame_idx]["reverse"]
        key = "cond_frame_outputs" if is_init else "non_cond_frame_outputs"
        obj_out, obj_tmp = st["output_dict_per_obj"][obj_idx], st["temp_output_dict_per_obj"][obj_idx]
        cur, _ = self.single_frame(st, obj_out, frame_idx, 1, is_init, None, m, reverse, False)
        obj_tmp[key][frame_idx] = cur
        cons = self.consolidate(st, frame_idx, is_init, False, True)
        return frame_idx, st["obj_ids"], self.video_res(st, cons["pred_masks_video_res"])

    def video_res(self, st, masks):
        """_get_orig_video_res_output (sam2_video_predictor.py:618-642)."""
        hw = (st["video_height"], st["video_width"])
        if masks.shape[-2:] == hw:
            return masks
        return F.interpolate(masks, size=hw, mode="bilinear", align_corners=False)

    # ------------------------------------------------------------------ consolidation (A9)
    def consolidate(self, st, frame_idx, is_cond, run_mem_encoder, at_video_res=False):
        """_consolidate_temp_output_across_obj (sam2_video_predictor.py:644-767)."""
        B = len(st["obj_idx_to_id"])
        key = "cond_frame_outputs" if is_cond else "non_cond_frame_outputs"
        if at_video_res:
            hw, mkey = (st["video_height"], st["video_width"]), "pred_masks_video_res"
        else:
            hw, mkey = (self.image_size // 4, self.image_size // 4), "pred_masks"
        cons = {"maskmem_features": None, "maskmem_pos_enc": None,
                mkey: torch.full((B, 1, *hw), NO_OBJ_SCORE),
                "obj_ptr": torch.full((B, self.hidden_dim), NO_OBJ_SCORE),
                "object_score_logits": torch.full((B, 1), 10.0)}
        empty_ptr = None
        for i in range(B):
            tmp, od = st["temp_output_dict_per_obj"][i], st["output_dict_per_obj"][i]
            out = tmp[key].get(frame_idx) or od["cond_frame_outputs"].get(frame_idx) \
                or od["non_cond_frame_outputs"].get(frame_idx)
            if out is None:
                if run_mem_encoder:
                    if empty_ptr is None:
                        empty_ptr = self.empty_mask_ptr(st, frame_idx)
                    cons["obj_ptr"][i:i + 1] = empty_ptr
                continue
            m = out["pred_masks"]
            cons[mkey][i:i + 1] = m if m.shape[-2:] == hw else F.interpolate(m, size=hw, mode="bilinear", align_corners=False)
            cons["obj_ptr"][i:i + 1] = out["obj_ptr"]
            cons["object_score_logits"][i:i + 1] = out["object_score_logits"]
        if run_mem_encoder:
            high = F.interpolate(cons["pred_masks"], size=(self.image_size, self.image_size), mode="bilinear",
                                 align_corners=False)
            feats, _ = self.image_feature(st, frame_idx, B)   # _run_memory_encoder :1367-1404
            f, pos = self.encode_new_memory(feats[-1], high, cons["object_score_logits"], True)
            cons["maskmem_features"] = f.to(torch.bfloat16)
            cons["maskmem_pos_enc"] = self.maskmem_pos_enc(st, pos)
        return cons

    def empty_mask_ptr(self, st, frame_idx):
        """_get_empty_mask_ptr (sam2_video_predictor.py:769-804)."""
        feats, poss = self.image_feature(st, frame_idx, 1)
        zeros = torch.zeros((1, 1, self.image_size, self.image_size))
        return self.track_step(frame_idx, True, feats, poss, None, zeros, {}, st["num_frames"], False, False)["obj_ptr"]

    def maskmem_pos_enc(self, st, pos):
        """_get_maskmem_pos_enc (sam2_video_predictor.py:1406-1435)."""
        if pos is None:
            return None
        c = st["constants"]
        if "maskmem_pos_enc" not in c:
            c["maskmem_pos_enc"] = [x[0:1].clone() for x in pos]
        return [x.expand(pos[0].size(0), -1, -1, -1) for x in c["maskmem_pos_enc"]]

    def add_output_per_object(self, st, frame_idx, out, key):
        """_add_output_per_object (sam2_video_predictor.py:1027-1058)."""
        for i, od in st["output_dict_per_obj"].items():
            s = slice(i, i + 1)
            o = {"maskmem_features": None, "maskmem_pos_enc": None, "pred_masks": out["pred_masks"][s],
                 "obj_ptr": out["obj_ptr"][s], "object_score_logits": out["object_score_logits"][s]}
            if out["maskmem_features"] is not None:
                o["maskmem_features"] = out["maskmem_features"][s]
            if out["maskmem_pos_enc"] is not None:
                o["maskmem_pos_enc"] = [x[s] for x in out["maskmem_pos_enc"]]
            od[key][frame_idx] = o

    # ------------------------------------------------------------------ prompt / object removal
    def reset_tracking_results(self, st):
        """_reset_tracking_results (sam2_video_predictor.py:1147-1172)."""
        for k in ("point_inputs_per_obj", "mask_inputs_per_obj"):
            for v in st[k].values():
                v.clear()
        for k in ("output_dict_per_obj", "temp_output_dict_per_obj"):
            for v in st[k].values():
                v["cond_frame_outputs"].clear()
                v["non_cond_frame_outputs"].clear()
        for k in ("output_dict", "consolidated_frame_inds"):
            st[k]["cond_frame_outputs"].clear()
            st[k]["non_cond_frame_outputs"].clear()
        st["tracking_has_started"] = False
        st["frames_already_tracked"].clear()

    def _masks_after_edit(self, st, frame_idx):
        is_cond = any(frame_idx in t["cond_frame_outputs"] for t in st["temp_output_dict_per_obj"].values())
        cons = self.consolidate(st, frame_idx, is_cond, False, at_video_res=True)
        return self.video_res(st, cons["pred_masks_video_res"])

    def clear_all_prompts_in_frame(self, st, frame_idx, obj_id, need_output=True):
        """clear_all_prompts_in_frame (sam2_video_predictor.py:1061-1131)."""
        i = self.obj_id_to_idx(st, obj_id)
        st["point_inputs_per_obj"][i].pop(frame_idx, None)
        st["mask_inputs_per_obj"][i].pop(frame_idx, None)
        st["temp_output_dict_per_obj"][i]["cond_frame_outputs"].pop(frame_idx, None)
        st["temp_output_dict_per_obj"][i]["non_cond_frame_outputs"].pop(frame_idx, None)
        B = len(st["obj_idx_to_id"])
        if not any(frame_idx in st["point_inputs_per_obj"][j] or frame_idx in st["mask_inputs_per_obj"][j] for j in range(B)):
            od, cfi = st["output_dict"], st["consolidated_frame_inds"]
            cfi["cond_frame_outputs"].discard(frame_idx)
            cfi["non_cond_frame_outputs"].discard(frame_idx)
            out = od["cond_frame_outputs"].pop(frame_idx, None)
            if out is not None:                                    # :1095-1099 demoted to a non-conditioning frame
                od["non_cond_frame_outputs"][frame_idx] = out
                st["frames_already_tracked"].pop(frame_idx, None)
            for j in range(B):
                o = st["output_dict_per_obj"][j]
                oo = o["cond_frame_outputs"].pop(frame_idx, None)
                if oo is not None:
                    o["non_cond_frame_outputs"][frame_idx] = oo
            if not od["cond_frame_outputs"]:                       # :1108-1110
                self.reset_tracking_results(st)
        if need_output:
            return frame_idx, st["obj_ids"], self._masks_after_edit(st, frame_idx)

    def remove_object(self, st, obj_id, strict=False, need_output=True):
        """remove_object (sam2_video_predictor.py:1438-1549)."""
        rm = st["obj_id_to_idx"].get(obj_id)
        if rm is None:
            if strict:
                raise RuntimeError(f"object id {obj_id} does not exist; existing ids: {st['obj_ids']}")
            return st["obj_ids"], []
        if len(st["obj_id_to_idx"]) == 1:                          # :1456-1459
            self.reset_tracking_results(st)
            for k in ("obj_id_to_idx", "obj_idx_to_id", "obj_ids", "point_inputs_per_obj", "mask_inputs_per_obj",
                      "output_dict_per_obj", "temp_output_dict_per_obj"):
                st[k].clear()
            return st["obj_ids"], []
        frames = set(st["point_inputs_per_obj"][rm]) | set(st["mask_inputs_per_obj"][rm])
        for t in frames:                                           # step 0 :1466-1476
            self.clear_all_prompts_in_frame(st, t, obj_id, need_output=False)
        old = list(range(len(st["obj_ids"])))
        remain = [i for i in old if i != rm]
        ids = [st["obj_ids"][i] for i in remain]                   # step 1 :1480-1491
        st["obj_id_to_idx"] = {o: n for n, o in enumerate(ids)}
        st["obj_idx_to_id"] = {n: o for n, o in enumerate(ids)}
        st["obj_ids"] = ids
        for k in ("point_inputs_per_obj", "mask_inputs_per_obj", "output_dict_per_obj", "temp_output_dict_per_obj"):
            moved = {remain.index(i): st[k].pop(i) for i in old if i != rm}    # step 2 :1493-1505
            st[k].pop(rm, None)
            st[k].update(moved)
        for key in ("cond_frame_outputs", "non_cond_frame_outputs"):           # step 3 :1507-1527
            for t, out in st["output_dict"][key].items():
                out["maskmem_features"] = out["maskmem_features"][remain]
                out["maskmem_pos_enc"] = self.maskmem_pos_enc(st, [x[remain] for x in out["maskmem_pos_enc"]])
                for f in ("pred_masks", "obj_ptr", "object_score_logits"):
                    out[f] = out[f][remain]
                self.add_output_per_object(st, t, out, key)
        updated = [(t, self._masks_after_edit(st, t)) for t in frames] if need_output else []   # step 4 :1531-1547
        return st["obj_ids"], updated

    # ------------------------------------------------------------------ propagate (A9, A10)
    def preflight(self, st):
        """propagate_in_video_preflight (sam2_video_predictor.py:807-893)."""
        st["tracking_has_started"] = True
        od, cfi = st["output_dict"], st["consolidated_frame_inds"]
        for is_cond in (False, True):
            key = "cond_frame_outputs" if is_cond else "non_cond_frame_outputs"
            inds = set()
            for tmp in st["temp_output_dict_per_obj"].values():
                inds.update(tmp[key].keys())
            cfi[key].update(inds)
            for t in inds:
                cons = self.consolidate(st, t, is_cond, True)
                od[key][t] = cons
                self.add_output_per_object(st, t, cons, key)
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

    def single_frame(self, st, output_dict, frame_idx, batch, is_init, point_inputs, mask_inputs, reverse,
                     run_mem_encoder, prev_sam_mask_logits=None):
        """_run_single_frame_inference (sam2_video_predictor.py:1280-1365), no hole filling."""
        feats, poss = self.image_feature(st, frame_idx, batch)
        cur = self.track_step(frame_idx, is_init, feats, poss, point_inputs, mask_inputs, output_dict,
                              st["num_frames"], reverse, run_mem_encoder, prev_sam_mask_logits,
                              st["preloading_memory_cond_frame_idx"])
        f = cur["maskmem_features"]
        if f is not None:
            f = f.to(torch.bfloat16)
        pm = cur["pred_masks"]
        if self.fill_hole_area > 0:   # sam2_video_predictor.py:1343-1346
            from .cc import fill_holes_in_mask_scores
            pm = torch.from_numpy(fill_holes_in_mask_scores(pm.numpy(), self.fill_hole_area))
        compact = {"maskmem_features": f, "maskmem_pos_enc": self.maskmem_pos_enc(st, cur["maskmem_pos_enc"]),
                   "pred_masks": pm, "obj_ptr": cur["obj_ptr"],
                   "object_score_logits": cur["object_score_logits"]}
        return compact, pm

    def propagate_in_video(self, st, start_frame_idx=None, max_frame_num_to_track=None, reverse=False):
        """propagate_in_video (sam2_video_predictor.py:911-1025); generator."""
        self.preflight(st)
        od, cfi = st["output_dict"], st["consolidated_frame_inds"]
        n, B = st["num_frames"], len(st["obj_idx_to_id"])
        if len(od["cond_frame_outputs"]) == 0:
            raise RuntimeError("no points are provided; please add points first")
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
        for t in order:
            if t in cfi["cond_frame_outputs"]:
                key = "cond_frame_outputs"
                cur = od[key][t]
                pm = cur["pred_masks"]
            elif t in cfi["non_cond_frame_outputs"]:
                key = "non_cond_frame_outputs"
                cur = od[key][t]
                pm = cur["pred_masks"]
            else:
                key = "non_cond_frame_outputs"
                cur, pm = self.single_frame(st, od, t, B, False, None, None, reverse, True)
                od[key][t] = cur
            self.add_output_per_object(st, t, cur, key)
            st["frames_already_tracked"][t] = {"reverse": reverse}
            yield t, st["obj_ids"], self.video_res(st, pm)

    def release_old_frames(self, st, frame_idx, max_frames, pre_frames, release_images=False):
        """release_old_frames (sam2_video_predictor.py:1215-1273)."""
        oldest = frame_idx - max_frames
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
        if release_images:
            old = [t for t in st["images_idx"] if pre_frames - 1 < t <= oldest]
            rm = {st["images_idx"].index(t) for t in old}
            keep = torch.tensor([i for i in range(st["images"].size(0)) if i not in rm])
            st["images"] = torch.index_select(st["images"], 0, keep)
            st["images_idx"] = [t for t in st["images_idx"] if t not in old]
