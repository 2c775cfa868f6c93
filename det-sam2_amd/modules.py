"""Module-level drop-ins: ``nn.Module``s with the SIGNATURES AND LAYOUTS of the reference's Hydra ``_target_`` modules
(SURVEY.md 8b "Module-level signatures a replacement must honour"), each backed by one PyTorch custom op (``torch.ops.det_sam2.*``) over one C-ABI stage of libdetsam2_hip.

    HipImageEncoder   ~ SAM2Base.forward_image            (sam2/modeling/sam2_base.py:450-461; backbones/image_encoder.py:30-43)
    HipMemoryAttention~ MemoryAttention.forward           (sam2/modeling/memory_attention.py:119-176)
    HipMemoryEncoder  ~ MemoryEncoder.forward             (sam2/modeling/memory_encoder.py:158-181)
    HipSamHeads       ~ SAM2Base._forward_sam_heads       (sam2/modeling/sam2_base.py:254-397) = PromptEncoder + MaskDecoder
    HipPromptEncoder  ~ PromptEncoder.forward / get_dense_pe / mask_input_size   (sam2/modeling/sam/prompt_encoder.py:134-171,64-71,49)
    HipMaskDecoder    ~ MaskDecoder.forward, .conv_s0 / .conv_s1                 (sam2/modeling/sam/mask_decoder.py:105-161,73-78)

They take and return the reference's tensors - ``[B,C,H,W]`` feature maps, ``(HW,B,C)`` token sequences - and convert to
the library's token-major layout at the boundary (permute + contiguous: PyTorch is plumbing here, every number comes
from the HIP kernels).  ``build_modules(cfg, state_dict)`` gives all four over ONE ``ds2_model``; a maintainer of the
reference assigns them over ``SAM2Base.image_encoder / memory_attention / memory_encoder``.
``tests/test_hip_modules.py`` feeds the seeded inputs of ``oracle/make_goldens.py:l1_inputs`` and compares with the
reference's own outputs (tests/golden/l1_*.npz) - HIP against the reference fixture, no oracle in between.
"""
from __future__ import annotations

import torch
from torch import nn

from .constants import sine_pos_2d
from .hip_model import TOK, HipSam2


def _tok(x):          # [B,C,H,W] -> token-major [B,H*W,C]
    return x.flatten(2).transpose(1, 2).contiguous()


def _map(x, h, w):    # token-major [B,H*W,C] -> [B,C,H,W]
    return x.transpose(1, 2).reshape(x.shape[0], x.shape[2], h, w)


class HipImageEncoder(nn.Module):
    """``forward(img_batch fp32 [N,3,1024,1024]) -> {"vision_features", "vision_pos_enc": [3], "backbone_fpn": [3]}`` with
    conv_s0 / conv_s1 already applied to levels 0 / 1, exactly what ``SAM2Base.forward_image`` returns."""

    def __init__(self, hip: HipSam2):
        super().__init__()
        self.hip = hip
        self._pos = None

    def _pos_enc(self, n):
        if self._pos is None:   # PositionEmbeddingSine is a constant of the geometry (position_encoding.py:79-112)
            self._pos = [_map(torch.from_numpy(sine_pos_2d(256, s, s)).to(self.hip.device)[None], s, s) for s in (256, 128, 64)]
        return [p.expand(n, -1, -1, -1) for p in self._pos]

    @torch.inference_mode()
    def forward(self, img_batch):
        h = self.hip
        x = img_batch.to(h.device, torch.float32).contiguous()
        n = x.shape[0]
        assert tuple(x.shape[1:]) == (3, h.cfg.image_size, h.cfg.image_size)
        f0, f1, f2 = h.ops.image_encoder(h._h, x)          # fp32 frames: forward_image's own input (sam2_video_predictor.py:1186)
        fpn = [_map(f0, 256, 256), _map(f1, 128, 128), _map(f2, 64, 64)]
        return {"vision_features": fpn[-1], "vision_pos_enc": self._pos_enc(n), "backbone_fpn": fpn}


class HipMemoryAttention(nn.Module):
    """``forward(curr, memory, curr_pos=None, memory_pos=None, num_obj_ptr_tokens=0) -> (HW,B,C)`` (sequence-first, as the
    reference passes them: sam2_base.py:676-683).  ``curr`` / ``curr_pos`` may be single-element lists."""

    def __init__(self, hip: HipSam2):
        super().__init__()
        self.hip = hip
        self.d_model = hip.cfg.d_model          # read by SAM2Base.__init__ (sam2_base.py:121)

    @torch.inference_mode()
    def forward(self, curr, memory, curr_pos=None, memory_pos=None, num_obj_ptr_tokens=0):
        h = self.hip
        if isinstance(curr, (list, tuple)):
            assert len(curr) == 1
            curr, curr_pos = curr[0], (curr_pos[0] if curr_pos is not None else None)
        assert curr.shape[0] == TOK and curr.shape[2] == 256 and memory.shape[2] == 64
        B, nk = curr.shape[1], memory.shape[0]
        c = curr.to(h.device, torch.float32).transpose(0, 1).contiguous()                 # [B,4096,256]
        cp = None if curr_pos is None else curr_pos.to(h.device, torch.float32).transpose(0, 1).contiguous()
        mem = memory.to(h.device, torch.float32).transpose(0, 1).contiguous()             # [B,Nk,64]
        mp = (torch.zeros_like(mem) if memory_pos is None else
              memory_pos.to(h.device, torch.float32).transpose(0, 1).contiguous())
        if cp is None:                                     # pos_enc_at_input with no positions given: zeros, shared
            cp = h._empty(TOK, 256).zero_()
        out = h.ops.memory_attention(h._h, B, c, cp, mem, mp, int(num_obj_ptr_tokens))
        return out.transpose(0, 1)


class HipMemoryEncoder(nn.Module):
    """``forward(pix_feat [B,256,64,64], masks [B,1,1024,1024], skip_mask_sigmoid=False) ->
    {"vision_features": [B,64,64,64], "vision_pos_enc": [pos]}``."""

    def __init__(self, hip: HipSam2):
        super().__init__()
        self.hip = hip
        self._pos = None

    @torch.inference_mode()
    def forward(self, pix_feat, masks, skip_mask_sigmoid=False):
        h = self.hip
        B = pix_feat.shape[0]
        pf = _tok(pix_feat.to(h.device, torch.float32))
        m = masks.to(h.device, torch.float32).reshape(B, h.cfg.image_size, h.cfg.image_size).contiguous()
        out = h.ops.memory_encoder_module(h._h, B, pf, m, bool(skip_mask_sigmoid))
        if self._pos is None:
            self._pos = _map(torch.from_numpy(sine_pos_2d(64, 64, 64)).to(h.device)[None], 64, 64)
        return {"vision_features": _map(out, 64, 64), "vision_pos_enc": [self._pos.expand(B, -1, -1, -1)]}


class HipSamHeads(nn.Module):
    """``forward(backbone_features [B,256,64,64], point_inputs=None, mask_inputs=None, high_res_features=None,
    multimask_output=False)`` -> the 7-tuple of ``_forward_sam_heads``.  The kernels select the output mask on the GPU, so
    the two multimask entries (low_res_multimasks, high_res_multimasks) hold the SELECTED mask only."""

    def __init__(self, hip: HipSam2):
        super().__init__()
        self.hip = hip

    @torch.inference_mode()
    def forward(self, backbone_features, point_inputs=None, mask_inputs=None, high_res_features=None, multimask_output=False):
        h = self.hip
        B = backbone_features.shape[0]
        pix = _tok(backbone_features.to(h.device, torch.float32))
        assert high_res_features is not None and len(high_res_features) == 2, "use_high_res_features_in_sam is on in SAM 2.1"
        # the stage shares the frame's high-res features between objects (they are .expand()-ed in the reference)
        f0 = _tok(high_res_features[0].to(h.device, torch.float32))
        f1 = _tok(high_res_features[1].to(h.device, torch.float32))
        coords = labels = None
        if point_inputs is not None:
            coords = point_inputs["point_coords"].to(h.device, torch.float32)
            labels = point_inputs["point_labels"].to(h.device, torch.int32)
        mi = None
        if mask_inputs is not None:
            mi = mask_inputs.to(h.device, torch.float32)
            if tuple(mi.shape[-2:]) != (256, 256):     # sam2_base.py:305-314
                mi = h.resize_aa(mi.reshape(B, *mi.shape[-2:]).contiguous(), 256, 256)
            mi = mi.reshape(B, 256, 256)
        lows, ptrs, objs, ious = [], [], [], []
        for b in range(B):        # per-object high-res features: one call per object (the tracking loop passes shared ones)
            low, ptr, obj, iou = h.sam_heads(1, pix[b:b + 1], f0[b], f1[b], None if coords is None else coords[b:b + 1],
                                             None if labels is None else labels[b:b + 1], bool(multimask_output),
                                             mask_inputs=None if mi is None else mi[b:b + 1])
            lows.append(low), ptrs.append(ptr), objs.append(obj), ious.append(iou)
        low = torch.cat(lows)[:, None]
        high, _ = h.mask_output(low[:, 0].contiguous(), h.cfg.image_size, h.cfg.image_size, want_logits=True, want_packed=False)
        obj = torch.cat(objs)[:, None]
        return low, high, torch.cat(ious)[:, None], low, high, torch.cat(ptrs), obj


class HipPromptEncoder(nn.Module):
    """``forward(points, boxes, masks) -> (sparse [B,N,256], dense [B,256,64,64])`` with the reference's argument meaning
    (prompt_encoder.py:134-171): ``points = (coords [B,P,2] in 1024-grid pixels, labels [B,P])`` - the padding point is
    appended when ``boxes is None`` (:81-85,158) -, ``boxes [B,4]`` become two corner points labelled 2 / 3 (:106-116),
    ``masks [B,1,256,256]`` go through ``mask_downscaling`` (:97-100), else ``no_mask_embed`` is broadcast (:167-169).
    ``get_dense_pe()`` (:64-71) and ``mask_input_size`` (:49) as in the reference."""

    def __init__(self, hip: HipSam2):
        super().__init__()
        self.hip = hip
        self.embed_dim = 256
        self.image_embedding_size = (64, 64)
        self.input_image_size = (hip.cfg.image_size, hip.cfg.image_size)
        self.mask_input_size = (256, 256)

    @torch.inference_mode()
    def get_dense_pe(self):
        return _map(self.hip.constant("#dense_pe").reshape(1, TOK, 256), 64, 64)

    @torch.inference_mode()
    def forward(self, points, boxes, masks):
        h = self.hip
        if points is not None:
            B = points[0].shape[0]
        elif boxes is not None:
            B = boxes.shape[0]
        elif masks is not None:
            B = masks.shape[0]
        else:
            B = 1
        coords = labels = None
        if points is not None:
            coords = points[0].to(h.device, torch.float32).reshape(B, -1, 2)
            labels = points[1].to(h.device, torch.int32).reshape(B, -1)
        if boxes is not None:
            bc = boxes.to(h.device, torch.float32).reshape(B, 2, 2)
            bl = torch.tensor([2, 3], dtype=torch.int32, device=h.device).expand(B, 2)
            coords = bc if coords is None else torch.cat([coords, bc], 1)
            labels = bl if labels is None else torch.cat([labels, bl], 1)
        mi = None if masks is None else masks.to(h.device, torch.float32).reshape(B, 256, 256).contiguous()
        sparse, dense = h.ops.prompt_encoder(h._h, B, None if coords is None else coords.contiguous(),
                                             None if labels is None else labels.contiguous(), boxes is None, mi, h.like)
        return sparse, _map(dense, 64, 64)


class _ConvS(nn.Module):
    """``conv_s0`` / ``conv_s1`` (mask_decoder.py:73-78) - folded into the image encoder stage here (its fpn levels 0 / 1 come
    out with them applied, as ``forward_image`` returns them, sam2_base.py:455-460): applied to a feature that still needs it
    through the reference's weights in fp32 on the device."""

    def __init__(self, hip, name):
        super().__init__()
        self.hip, self.name = hip, name

    @torch.inference_mode()
    def forward(self, x):
        w = self.hip.parameter(f"sam_mask_decoder.{self.name}.weight")
        b = self.hip.parameter(f"sam_mask_decoder.{self.name}.bias")
        return torch.nn.functional.conv2d(x.to(self.hip.device, torch.float32), w, b)


class HipMaskDecoder(nn.Module):
    """``forward(image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output, repeat_image,
    high_res_features=None) -> (masks, iou_pred, sam_tokens_out, object_score_logits)`` (mask_decoder.py:105-161).  The
    two-way transformer, the upscaling and the hypernetwork / IoU / object-score heads run in ``ds2_mask_decoder``; the output
    slicing of ``forward`` (multimask slice, ``_dynamic_multimask_via_stability`` :261-296) is index arithmetic on four
    masks per object, done here on the device."""

    def __init__(self, hip: HipSam2):
        super().__init__()
        self.hip = hip
        self.conv_s0, self.conv_s1 = _ConvS(hip, "conv_s0"), _ConvS(hip, "conv_s1")
        self.dynamic_multimask_via_stability = hip.cfg.dynamic_multimask_via_stability        # build_sam.py:126-135
        self.dynamic_multimask_stability_delta = hip.cfg.dynamic_multimask_stability_delta
        self.dynamic_multimask_stability_thresh = hip.cfg.dynamic_multimask_stability_thresh
        self.use_multimask_token_for_obj_ptr = True

    def _dynamic_multimask_via_stability(self, masks, iou):
        B = masks.shape[0]
        ar = torch.arange(B, device=masks.device)
        best = torch.argmax(iou[:, 1:], dim=-1)
        best_masks, best_iou = masks[:, 1:][ar, best][:, None], iou[:, 1:][ar, best][:, None]
        single, single_iou = masks[:, 0:1], iou[:, 0:1]
        flat = single.flatten(-2)
        d = self.dynamic_multimask_stability_delta
        area_i, area_u = (flat > d).sum(-1).float(), (flat > -d).sum(-1).float()
        stable = torch.where(area_u > 0, area_i / area_u, torch.ones_like(area_u)) >= self.dynamic_multimask_stability_thresh
        return (torch.where(stable[..., None, None].expand_as(single), single, best_masks),
                torch.where(stable.expand_as(single_iou), single_iou, best_iou))

    @torch.inference_mode()
    def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output,
                repeat_image, high_res_features=None):
        h = self.hip
        assert high_res_features is not None and len(high_res_features) == 2, "use_high_res_features is on in SAM 2.1"
        sp = sparse_prompt_embeddings.to(h.device, torch.float32).contiguous()
        B = sp.shape[0]
        emb = image_embeddings.to(h.device, torch.float32)
        if repeat_image:
            emb = torch.repeat_interleave(emb, B, dim=0)
        assert emb.shape[0] == B
        emb, dense = _tok(emb), _tok(dense_prompt_embeddings.to(h.device, torch.float32).expand(B, -1, -1, -1))
        pe = _tok(image_pe.to(h.device, torch.float32))[0].contiguous()
        f0 = _tok(high_res_features[0].to(h.device, torch.float32).expand(B, -1, -1, -1))
        f1 = _tok(high_res_features[1].to(h.device, torch.float32).expand(B, -1, -1, -1))
        outs = [h.ops.mask_decoder(h._h, 1, emb[b:b + 1], pe, sp[b:b + 1], dense[b:b + 1], f0[b], f1[b]) for b in range(B)]
        masks, iou, tok, obj = (torch.cat([o[i] for o in outs]) for i in range(4))
        if multimask_output:
            masks, iou = masks[:, 1:], iou[:, 1:]
        elif self.dynamic_multimask_via_stability:
            masks, iou = self._dynamic_multimask_via_stability(masks, iou)
        else:
            masks, iou = masks[:, 0:1], iou[:, 0:1]
        tokens = tok[:, 1:] if (multimask_output and self.use_multimask_token_for_obj_ptr) else tok[:, 0:1]
        return masks, iou, tokens, obj[:, None]


def build_modules(cfg, state_dict, device="cuda:0", max_batch=16):
    """-> dict(image_encoder, memory_attention, memory_encoder, sam_heads, sam_prompt_encoder, sam_mask_decoder, hip) over one
    shared ``ds2_model``."""
    hip = HipSam2(cfg, state_dict, device, max_batch)
    return {"hip": hip, "image_encoder": HipImageEncoder(hip), "memory_attention": HipMemoryAttention(hip),
            "memory_encoder": HipMemoryEncoder(hip), "sam_heads": HipSamHeads(hip),
            "sam_prompt_encoder": HipPromptEncoder(hip), "sam_mask_decoder": HipMaskDecoder(hip)}
