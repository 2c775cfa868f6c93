"""ctypes binding of libdetsam2_hip.so (C-ABI: include/detsam2_hip.h).

There is NO CPU fallback: if the library is missing, or a call fails, this raises.  Build it with
``python -c 'import __graft_entry__ as g; g.build()'`` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DS2_LIB: alternative build of the same library (A/B timing of kernel variants inside ONE gpurun call, see tools/ab.py)
LIB_PATH = os.environ.get("DS2_LIB") or os.path.join(_HERE, "lib", "libdetsam2_hip.so")

c_f32p = C.c_void_p
c_vp = C.c_void_p
i32 = C.c_int32


class Ds2Config(C.Structure):
    _fields_ = [
        ("image_size", i32), ("embed_dim", i32), ("num_heads", i32), ("stages", i32 * 4),
        ("global_att_blocks", i32 * 4), ("n_global_att_blocks", i32), ("window_spec", i32 * 4),
        ("d_model", i32), ("mem_dim", i32), ("num_maskmem", i32), ("mem_attn_layers", i32),
        ("mem_attn_ffn", i32), ("max_batch", i32),
        ("sigmoid_scale_for_mem_enc", C.c_float), ("sigmoid_bias_for_mem_enc", C.c_float),
        ("dynamic_multimask_stability_delta", C.c_float), ("dynamic_multimask_stability_thresh", C.c_float),
    ]


# name -> (restype, argtypes); must list every symbol include/detsam2_hip.h declares
SIGNATURES = {
    "ds2_last_error": (C.c_char_p, []),
    "ds2_abi_version": (C.c_int, []),
    "ds2_model_create": (C.c_int, [C.POINTER(Ds2Config), C.POINTER(c_vp)]),
    "ds2_model_destroy": (None, [c_vp]),
    "ds2_model_set_param": (C.c_int, [c_vp, C.c_char_p, c_vp, C.c_int64]),
    "ds2_model_finalize": (C.c_int, [c_vp, c_vp]),
    "ds2_model_create_view": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "ds2_ingest_frames": (C.c_int, [c_vp, c_vp, i32, i32, i32, c_vp, c_vp]),
    "ds2_image_encoder": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ds2_image_encoder_batch": (C.c_int, [c_vp, c_vp, i32, c_vp, c_vp, c_vp, c_vp]),
    "ds2_bank_assemble": (C.c_int, [c_vp, i32, i32, C.POINTER(c_vp), C.POINTER(i32), i32, C.POINTER(c_vp),
                                    C.POINTER(C.c_float), c_vp, c_vp, c_vp]),
    "ds2_memory_attention": (C.c_int, [c_vp, i32, c_vp, c_vp, c_vp, i32, i32, c_vp, c_vp]),
    "ds2_op_query_fragments": (C.c_int, [c_vp, i32, c_vp, i32, i32, c_vp, c_vp]),
    "ds2_bank_memory_attention": (C.c_int, [c_vp, i32, c_vp, i32, C.POINTER(c_vp), C.POINTER(i32), i32, C.POINTER(c_vp),
                                            C.POINTER(C.c_float), c_vp, c_vp]),
    "ds2_sam_heads": (C.c_int, [c_vp, i32, c_vp, i32, i32, c_vp, c_vp, c_vp, c_vp, i32, i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ds2_sam_heads_mask": (C.c_int, [c_vp, i32, c_vp, i32, i32, c_vp, c_vp, c_vp, c_vp, i32, c_vp, i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ds2_prompt_encoder": (C.c_int, [c_vp, i32, c_vp, c_vp, i32, i32, c_vp, c_vp, c_vp, c_vp]),
    "ds2_mask_decoder": (C.c_int, [c_vp, i32, c_vp, c_vp, c_vp, i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ds2_memory_encoder": (C.c_int, [c_vp, i32, c_vp, c_vp, c_vp, i32, c_vp, c_vp]),
    "ds2_connected_components": (C.c_int, [c_vp, i32, i32, i32, c_vp, c_vp, c_vp, c_vp]),
    "ds2_fill_holes": (C.c_int, [c_vp, i32, i32, i32, i32, c_vp, c_vp]),
    "ds2_yolo_postprocess_work_bytes": (C.c_int64, [i32, i32]),
    "ds2_yolo_postprocess": (C.c_int, [c_vp, i32, i32, i32, C.c_float, C.c_float, i32, c_vp, c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "ds2_image_encoder_f32": (C.c_int, [c_vp, c_vp, i32, c_vp, c_vp, c_vp, c_vp]),
    "ds2_memory_attention_ex": (C.c_int, [c_vp, i32, c_vp, i32, c_vp, i32, c_vp, c_vp, i32, i32, c_vp, c_vp]),
    "ds2_memory_encoder_ex": (C.c_int, [c_vp, i32, c_vp, i32, c_vp, i32, c_vp, c_vp]),
    "ds2_resize_aa": (C.c_int, [c_vp, i32, i32, i32, i32, i32, C.c_float, C.c_float, C.c_float, c_vp, c_vp, c_vp]),
    "ds2_mask_prompt_prepare": (C.c_int, [c_vp, i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ds2_obj_ptr_gate": (C.c_int, [c_vp, i32, c_vp, c_vp, c_vp]),
    "ds2_mask_output": (C.c_int, [c_vp, c_vp, i32, i32, i32, c_vp, c_vp, c_vp]),
    "ds2_set_precision": (C.c_int, [i32]),
    "ds2_get_precision": (C.c_int, []),
    "ds2_model_set_precision": (C.c_int, [c_vp, i32]),
    "ds2_model_get_precision": (C.c_int, [c_vp]),
    "ds2_profile_enable": (C.c_int, [i32]),
    "ds2_profile_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ds2_profile_tags": (C.c_int, [C.c_char_p, C.c_int64]),
    "ds2_op_gemm": (C.c_int, [i32, i32, i32, c_vp, i32, c_vp, i32, c_vp, c_vp, i32, i32, c_vp, c_vp, i32, i32, c_vp]),
    "ds2_op_gemm_planes": (C.c_int, [i32, i32, i32, c_vp, i32, c_vp, i32, c_vp, i32, c_vp, c_vp, c_vp]),
    "ds2_op_split_planes": (C.c_int, [c_vp, i32, i32, i32, i32, c_vp, c_vp, c_vp]),
    "ds2_op_linear_small": (C.c_int, [i32, i32, i32, c_vp, i32, c_vp, i32, c_vp, c_vp, i32, i32, c_vp, c_vp, i32, i32, c_vp]),
    "ds2_op_mlp": (C.c_int, [i32, i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, i32, c_vp]),
    "ds2_op_layernorm": (C.c_int, [c_vp, c_vp, c_vp, c_vp, i32, i32, C.c_float, i32, c_vp]),
    "ds2_op_attention": (C.c_int, [c_vp, c_vp, c_vp, c_vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.c_float,
                                   i32, i32, i32, i32, i32, i32, i32, c_vp, c_vp, c_vp]),
}

# the PyTorch custom ops csrc/torch_ops.cpp registers (torch.ops.det_sam2.<name>)
TORCH_OPS = ("ingest_frames", "image_encoder", "bank_assemble", "memory_attention", "sam_heads", "prompt_encoder", "mask_decoder", "memory_encoder",
             "memory_encoder_module", "resize_aa", "mask_prompt_prepare", "obj_ptr_gate", "mask_output",
             "get_connected_componnets", "fill_holes", "yolo_postprocess")

_lib = None
_ops = None
# (an A/B build selected with DS2_LIB=<dir>/ab_X.so brings its own op library <dir>/ab_X_torch.so, linked against it:
# tools/ab.py build)
TORCH_LIB_PATH = (LIB_PATH[:-3] + "_torch.so") if os.environ.get("DS2_LIB") else os.path.join(os.path.dirname(LIB_PATH), "libdetsam2_torch.so")


def load_torch_ops():
    """Register the PyTorch custom ops (csrc/torch_ops.cpp: TORCH_LIBRARY(det_sam2, m) over the C-ABI) and return
    ``torch.ops.det_sam2``.  Raises if the library is not built - the stages have no other way in."""
    global _ops
    if _ops is not None:
        return _ops
    import torch
    load()                                       # libdetsam2_hip.so first (the op library links against it)
    if not os.path.exists(TORCH_LIB_PATH):
        raise ImportError(f"{TORCH_LIB_PATH} is missing: build it with: python -c 'import __graft_entry__ as g; g.build()'")
    torch.ops.load_library(TORCH_LIB_PATH)
    _ops = torch.ops.det_sam2
    return _ops


class Ds2Error(RuntimeError):
    pass


def load():
    """Load the library (once) and attach signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is required (no CPU fallback exists). "
            "Build it with: python -c 'import __graft_entry__ as g; g.build()'")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.ds2_abi_version() != 1:
        raise ImportError("libdetsam2_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ds2_last_error()
        raise Ds2Error(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
