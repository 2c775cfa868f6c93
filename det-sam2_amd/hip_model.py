"""SAM 2.1 stages on MI355X: thin host wrapper over the C-ABI (include/detsam2_hip.h).

The STAGES are called as PyTorch-ROCm custom ops (``torch.ops.det_sam2.*``, csrc/torch_ops.cpp: TORCH_LIBRARY over the
C-ABI - tensors in, tensors out, current HIP stream taken inside the op); model lifetime, parameters, precision,
profiling and the primitive test ops go through ctypes.  PyTorch is plumbing only (allocations, streams); every number is
produced by the kernels in csrc/; nothing falls back to ATen arithmetic.

Token-major layout: what the reference holds as [B,C,H,W] is [B,H*W,C] here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from .config import ModelCfg, resolve_config
from .constants import model_constants
from .weights import check_state_dict

TOK = 4096


def _p(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


class HipOps:
    """Library handle + stream/allocation plumbing + the primitive ops (no model needed)."""

    def __init__(self, device="cuda:0"):
        self.lib = _capi.load()                     # raises if the HIP library is not built
        self.ops = _capi.load_torch_ops()           # torch.ops.det_sam2 (raises if the op library is not built)
        if not torch.cuda.is_available():
            raise RuntimeError("det-sam2_amd needs a ROCm GPU: there is no CPU execution path")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    PRECISIONS = ("fp32", "bf16x3", "bf16x3k")

    def set_precision(self, mode: str):
        """'fp32' (exact fp32 MFMA), 'bf16x3' (split-precision bf16 MFMA: three product terms everywhere) or 'bf16x3k'
        (default: bf16x3, with the memory-attention scores as plain bf16 x bf16 products and its softmax weights as one bf16 plane).
        On a model (HipSam2) this sets THAT model's mode; on a bare HipOps the process default used by the primitive ops
        and by models created afterwards."""
        if mode not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {self.PRECISIONS}, got {mode!r}")
        h = getattr(self, "h", None)
        if h:
            _capi.check(self.lib.ds2_model_set_precision(h, self.PRECISIONS.index(mode)), "ds2_model_set_precision")
        else:
            _capi.check(self.lib.ds2_set_precision(self.PRECISIONS.index(mode)), "ds2_set_precision")

    def get_precision(self) -> str:
        h = getattr(self, "h", None)
        return self.PRECISIONS[self.lib.ds2_model_get_precision(h) if h else self.lib.ds2_get_precision()]

    # ------------------------------------------------------------------ measurement
    def profile_enable(self, on=True, gemm_shapes=False):
        """HIP-event brackets around the stages / dominant kernels; gemm_shapes: also one bracket per GEMM ("gemm M N K")."""
        _capi.check(self.lib.ds2_profile_enable(2 if (on and gemm_shapes) else int(bool(on))), "ds2_profile_enable")

    def profile_read(self, tag):
        """-> (total_ms, launches) of the HIP-event brackets recorded under ``tag`` since the last read."""
        ms, n = C.c_double(0), C.c_int64(0)
        _capi.check(self.lib.ds2_profile_read(tag.encode(), C.byref(ms), C.byref(n)), "ds2_profile_read")
        return ms.value, n.value

    def profile_tags(self):
        buf = C.create_string_buffer(1 << 16)
        _capi.check(self.lib.ds2_profile_tags(buf, len(buf)), "ds2_profile_tags")
        return [t for t in buf.value.decode().split("\n") if t]

    # ------------------------------------------------------------------ primitives (tests)
    def op_gemm(self, A, W, bias=None, act=0, gamma=None, R=None, r_mod=0):
        M, K = A.shape
        N = W.shape[0]
        out = self._empty(M, N)
        _capi.check(self.lib.ds2_op_gemm(M, N, K, _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(out), N, act, _p(gamma),
                                         _p(R), 0 if R is None else R.stride(0), r_mod, self._stream()), "ds2_op_gemm")
        return out

    def op_gemm_planes(self, A, W, bias=None, act=0):
        """act(A W^T + bias) as bf16 operand planes (ds2_op_gemm_planes): -> (hi, lo) int16 tensors [M, round32(N)]."""
        import torch
        M, K = A.shape
        N = W.shape[0]
        ld = (N + 31) // 32 * 32
        hi = torch.empty(M, ld, dtype=torch.int16, device=self.device)
        lo = torch.empty(M, ld, dtype=torch.int16, device=self.device)
        _capi.check(self.lib.ds2_op_gemm_planes(M, N, K, _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), act, _p(hi), _p(lo),
                                                self._stream()), "ds2_op_gemm_planes")
        return hi, lo

    def op_split_planes(self, x, fmt):
        """The operand planes of x [rows, cols] (ds2_op_split_planes): fmt 0 bf16 hi / lo, 1 MX activation, 2 MX weight, 3 fp16 hi / lo
        -> (p1, p2) int16 tensors [rows, round32(cols)]."""
        import torch
        rows, cols = x.shape
        ld = (cols + 31) // 32 * 32
        p1 = torch.empty(rows, ld, dtype=torch.int16, device=self.device)
        p2 = torch.empty(rows, ld, dtype=torch.int16, device=self.device)
        _capi.check(self.lib.ds2_op_split_planes(_p(x), x.stride(0), rows, cols, int(fmt), _p(p1), _p(p2), self._stream()), "ds2_op_split_planes")
        return p1, p2

    def op_linear_small(self, A, W, bias=None, act=0, gamma=None, R=None, r_mod=0):
        """Few-row Linear layer in exact fp32 (ds2_op_linear_small): A [M<=128,K], W [N,K] -> [M,N]."""
        M, K = A.shape
        N = W.shape[0]
        out = self._empty(M, N)
        _capi.check(self.lib.ds2_op_linear_small(M, N, K, _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(out), N, act,
                                                 _p(gamma), _p(R), 0 if R is None else R.stride(0), r_mod, self._stream()),
                    "ds2_op_linear_small")
        return out

    def op_mlp(self, X, W1, b1, W2, b2, gamma=None, R=None, act=1):
        """Fused two-layer MLP of width 256 (ds2_op_mlp): X [rows,256], W1 [H,256], W2 [256,H] -> [rows,256]."""
        rows, H = X.shape[0], W1.shape[0]
        out = self._empty(rows, 256)
        _capi.check(self.lib.ds2_op_mlp(rows, H, _p(X), _p(W1), _p(b1), _p(W2), _p(b2), _p(gamma), _p(R), _p(out), act, self._stream()),
                    "ds2_op_mlp")
        return out

    def op_layernorm(self, x, w, b, eps, act=0):
        out = torch.empty_like(x)
        _capi.check(self.lib.ds2_op_layernorm(_p(x), _p(w), _p(b), _p(out), x.shape[0], x.shape[1], eps, act, self._stream()),
                    "ds2_op_layernorm")
        return out

    def op_attention(self, q, k, v, heads, scale, win_q=0, win_k=0, hq=0, wq=0, hk=0, wk=0, nwx=0, k_pad=None, v_pad=None,
                     batch=None, lq=None, lk=None, dv=None):
        """Plain mode: q [B,Lq,H*D], k [B,Lk,H*D], v [B,Lk,H*DV]. Windowed: q [Hq*Wq,H*D], k/v [Hk*Wk,...]."""
        D = q.shape[-1] // heads
        DV = dv if dv is not None else v.shape[-1] // heads
        if win_k == 0:
            batch, lq, lk = q.shape[0], q.shape[1], k.shape[1]
            o = self._empty(batch, lq, heads * DV)
        else:
            o = torch.zeros(q.shape[0], heads * DV, device=self.device)
        _capi.check(self.lib.ds2_op_attention(_p(q), _p(k), _p(v), _p(o), q.stride(-2), k.stride(-2), v.stride(-2), heads * DV,
                                              batch, heads, D, DV, lq, lk, scale, win_q, win_k, hq, wq, hk, wk, nwx, _p(k_pad),
                                              _p(v_pad), self._stream()), "ds2_op_attention")
        return o


class HipSam2(HipOps):
    """Owns one ``ds2_model`` (weights + workspace) on one GPU."""

    def __init__(self, cfg, state_dict, device="cuda:0", max_batch: int = 16):
        super().__init__(device)
        self.cfg: ModelCfg = resolve_config(cfg)
        check_state_dict(self.cfg, state_dict)      # strict, like build_sam.py:166-177
        t = self.cfg.trunk
        c = _capi.Ds2Config()
        c.image_size, c.embed_dim, c.num_heads = self.cfg.image_size, t.embed_dim, t.num_heads
        for i in range(4):
            c.stages[i] = t.stages[i]
            c.window_spec[i] = t.window_spec[i]
            c.global_att_blocks[i] = t.global_att_blocks[i] if i < len(t.global_att_blocks) else -1
        c.n_global_att_blocks = len(t.global_att_blocks)
        c.d_model, c.mem_dim, c.num_maskmem = self.cfg.d_model, self.cfg.mem_dim, self.cfg.num_maskmem
        c.mem_attn_layers, c.mem_attn_ffn, c.max_batch = self.cfg.mem_attn_layers, self.cfg.mem_attn_ffn, max_batch
        c.sigmoid_scale_for_mem_enc, c.sigmoid_bias_for_mem_enc = (self.cfg.sigmoid_scale_for_mem_enc,
                                                                   self.cfg.sigmoid_bias_for_mem_enc)
        c.dynamic_multimask_stability_delta = self.cfg.dynamic_multimask_stability_delta
        # (via_stability off: the single-mask output is always token 0 = "always stable")
        c.dynamic_multimask_stability_thresh = (self.cfg.dynamic_multimask_stability_thresh
                                                if self.cfg.dynamic_multimask_via_stability else float("-inf"))
        h = C.c_void_p()
        _capi.check(self.lib.ds2_model_create(C.byref(c), C.byref(h)), "ds2_model_create")
        self.h = h
        sd_np = {}
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            sd_np[k] = np.ascontiguousarray(a, dtype=np.float32)
        consts = model_constants(self.cfg, sd_np)
        for k, a in list(sd_np.items()) + list(consts.items()):
            a = np.ascontiguousarray(a)
            _capi.check(self.lib.ds2_model_set_param(self.h, k.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes),
                        f"set_param({k})")
        _capi.check(self.lib.ds2_model_finalize(self.h, self._stream()), "ds2_model_finalize")
        self.no_obj_ptr = torch.from_numpy(sd_np["no_obj_ptr"]).to(self.device)           # [1,256]
        # the few tensors the module-level drop-ins hand out themselves (modules.HipPromptEncoder.get_dense_pe, .conv_s0/.conv_s1)
        keep = ["#dense_pe"] + [f"sam_mask_decoder.conv_s{i}.{w}" for i in (0, 1) for w in ("weight", "bias")]
        self._kept = {k: torch.from_numpy(np.array((consts if k in consts else sd_np)[k], dtype=np.float32)).to(self.device) for k in keep}
        self.like = self.no_obj_ptr                                                        # a device tensor for op dispatch

    def constant(self, name):
        return self._kept[name]

    def parameter(self, name):
        return self._kept[name]

    @classmethod
    def view_of(cls, parent: "HipSam2") -> "HipSam2":
        """A second execution context over ``parent``'s weights (ds2_model_create_view): no parameter copies, no second set
        of weight planes - its own workspace arena only.  Keeps ``parent`` alive."""
        self = cls.__new__(cls)
        HipOps.__init__(self, parent.device)
        self.cfg, self._parent = parent.cfg, parent
        h = C.c_void_p()
        _capi.check(self.lib.ds2_model_create_view(parent.h, C.byref(h)), "ds2_model_create_view")
        self.h = h
        self.no_obj_ptr, self._kept, self.like = parent.no_obj_ptr, parent._kept, parent.like
        return self

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ds2_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ stages (torch.ops.det_sam2.*)
    @property
    def _h(self):
        return int(self.h.value)

    def ingest(self, frames_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [n,H,W,3] (device) -> fp16 [n,3,S,S] normalised frames (A3)."""
        return self.ops.ingest_frames(self._h, frames_u8)

    def image_encoder(self, frame_f16: torch.Tensor):
        """fp16 [3,S,S] -> (fpn0 [65536,32], fpn1 [16384,64], fpn2 [4096,256]) (A4+A5)."""
        f0, f1, f2 = self.ops.image_encoder(self._h, frame_f16[None])
        return f0[0], f1[0], f2[0]

    def image_encoder_batch(self, frames_f16: torch.Tensor):
        """fp16 [n,3,S,S] -> list of n (fpn0, fpn1, fpn2) tuples (views of three batched buffers)."""
        f0, f1, f2 = self.ops.image_encoder(self._h, frames_f16)
        return [(f0[i], f1[i], f2[i]) for i in range(frames_f16.shape[0])]

    def bank_assemble(self, B, mem_entries, ptr_entries):
        """mem_entries: [(bf16 [B,4096,64], tpos_row)], ptr_entries: [(fp32 [B,256], pos/t_diff_max)] (A11)
        -> memory, memory_pos fp32 [B,Nk,64]."""
        return self.ops.bank_assemble(self._h, B, [t for t, _ in mem_entries], [int(r) for _, r in mem_entries],
                                      [t for t, _ in ptr_entries], [float(p) for _, p in ptr_entries])

    def op_query_fragments(self, layer, x, fused):
        """test hook (ds2_op_query_fragments): x fp32 [rows,256] -> the layer's cross-attention queries as fp16 Q fragments [rows*256] int16."""
        import torch
        out = torch.empty(x.shape[0] * 256, dtype=torch.int16, device=x.device)
        _capi.check(self.lib.ds2_op_query_fragments(self.h, int(layer), C.c_void_p(x.data_ptr()), x.shape[0], int(bool(fused)),
                                                    C.c_void_p(out.data_ptr()), self._stream()), "ds2_op_query_fragments")
        return out

    def bank_attention(self, B, curr, mem_entries, ptr_entries):
        """bank_assemble + memory_attention in one call (the tracking loop; A11 + A12): same result bit for bit; in mode bf16x3k the bank's
        entries become the cross-attention's operands directly - the fp32 memory / memory_pos tensors are never written."""
        return self.ops.bank_memory_attention(self._h, B, curr, [t for t, _ in mem_entries], [int(r) for _, r in mem_entries],
                                              [t for t, _ in ptr_entries], [float(p) for _, p in ptr_entries])

    def memory_attention(self, B, curr, memory, memory_pos, num_obj_ptr_tokens):
        """curr [4096,256] shared; memory/memory_pos [B,Nk,64] -> [B,4096,256] (A12)."""
        return self.ops.memory_attention(self._h, B, curr, None, memory, memory_pos, num_obj_ptr_tokens)

    def sam_heads(self, B, pix_feat, fpn0, fpn1, point_coords=None, point_labels=None, multimask=False,
                  pix_bcast=False, add_no_mem_embed=False, mask_inputs=None):
        """-> low_res [B,256,256], obj_ptr [B,256], obj_logits [B], ious [B] (A7+A8).  mask_inputs [B,256,256]:
        optional mask prompt (logits) for the prompt encoder's dense embedding."""
        if point_coords is not None and point_coords.shape[1] == 0:
            point_coords = point_labels = None
        if point_coords is not None:
            point_coords, point_labels = point_coords.contiguous(), point_labels.contiguous()
        if mask_inputs is not None:
            mask_inputs = mask_inputs.contiguous()
        return self.ops.sam_heads(self._h, B, pix_feat, bool(pix_bcast), bool(add_no_mem_embed), fpn0, fpn1, point_coords,
                                  point_labels, mask_inputs, bool(multimask))

    def resize_aa(self, x, hout, wout, in_scale=1.0, in_bias=0.0, threshold=float("inf")):
        """F.interpolate(bilinear, antialias=True, align_corners=False) of fp32 [B,Hin,Win] -> [B,hout,wout]."""
        return self.ops.resize_aa(x, hout, wout, float(in_scale), float(in_bias), float(threshold))

    def use_mask_as_output(self, B, fpn2, fpn0, fpn1, mask):
        """SAM2Base._use_mask_as_output (sam2_base.py:399-448): mask fp32 0/1 [B,S,S] ->
        (low_res logits [B,256,256], obj_ptr [B,256], obj_logits [B]).  fpn2 is the frame's raw level-2 feature
        (no memory, no no_mem_embed: track_step :873-879)."""
        S = self.cfg.image_size
        assert mask.dtype == torch.float32 and tuple(mask.shape) == (B, S, S) and mask.is_contiguous()
        low = self.resize_aa(mask, S // 4, S // 4, 20.0, -10.0)                        # (mask*20-10) -> 256^2, antialiased
        ds, obj = self.ops.mask_prompt_prepare(self._h, mask)
        _, ptr, _, _ = self.sam_heads(B, fpn2, fpn0, fpn1, None, None, multimask=False, pix_bcast=True, add_no_mem_embed=False,
                                      mask_inputs=ds)                                  # gated by the decoder's own object score
        return low, self.ops.obj_ptr_gate(self._h, ptr, obj), obj                      # ... then by the mask's

    def memory_encoder(self, B, fpn2, low_res, obj_logits, binarize):
        """-> maskmem bf16 [B,4096,64] (A13)."""
        return self.ops.memory_encoder(self._h, B, fpn2, low_res, obj_logits, bool(binarize))

    def mask_output(self, low_res, hv, wv, want_logits=True, want_packed=True):
        """low_res [B,256,256] -> (logits fp32 [B,1,hv,wv] | None, packed uint8 [B,hv,ceil(wv/8)] | None) (A15);
        packed rows are numpy.packbits rows (MSB first, last byte zero-padded)."""
        logits, packed = self.ops.mask_output(self._h, low_res, hv, wv, bool(want_logits), bool(want_packed))
        return (logits if want_logits else None), (packed if want_packed else None)
