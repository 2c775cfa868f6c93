"""Preload memory-bank files.

The reference saves a bank by pickling the WHOLE inference state - frames, device objects, per-object views
(det_sam2_RT.py:489-497) - and loads it with ``pickle.load`` (:499-503).  That file is device-bound, unsafe to load
and ~6 MiB per frame larger than it needs to be.  ``DS2BANK`` is the replacement behind the same two methods
(``VideoProcessor.save_inference_state / load_inference_state``):

    bytes 0..7    magic  b"DS2BANK1"
    bytes 8..15   little-endian uint64: length L of the JSON header
    bytes 16..    JSON header (utf-8), then zero padding to a multiple of 64
    then          raw little-endian tensor payloads, each 64-byte aligned, at the offsets the header names

Header: ``{"version": 1, "model": cfg name, "layout": "token-major", "num_frames", "video_height", "video_width",
"obj_ids": [...], "entries": [{"frame": t, "kind": "cond" | "non_cond", "tensors": {name: {"dtype", "shape",
"offset", "nbytes"}}}]}``.  Per entry: ``maskmem_features`` bf16 [B,4096,64] (token-major; the reference holds
[B,64,64,64]), ``pred_masks`` f32 [B,1,256,256], ``obj_ptr`` f32 [B,256], ``object_score_logits`` f32 [B,1] and, for
conditioning frames, ``fpn2`` f32 [4096,256]: the frame's level-2 image feature, the only thing the online
new-object path (A17, sam2_video_predictor.py:281-310) needs from a preload frame - so NO frames are stored.

``load_bank`` also reads a reference-produced pickle (through an unpickler restricted to tensors / containers) and
converts it to the same in-memory form, so banks written by the reference stay usable.
"""
from __future__ import annotations

import io
import json
import os
import pickle
import struct
from collections import OrderedDict

import numpy as np
import torch

MAGIC = b"DS2BANK1"
_DT = {"bf16": (torch.bfloat16, 2), "f32": (torch.float32, 4), "f16": (torch.float16, 2)}
_NAME = {torch.bfloat16: "bf16", torch.float32: "f32", torch.float16: "f16"}
ENTRY_TENSORS = ("maskmem_features", "pred_masks", "obj_ptr", "object_score_logits")


def _raw(t: torch.Tensor) -> bytes:
    t = t.detach().contiguous().cpu()
    return t.view(torch.uint8).numpy().tobytes() if t.numel() else b""


def save_bank(path, inference_state, model_name: str, fpn2_of=None) -> dict:
    """Write the conditioning / non-conditioning entries and the object table of ``inference_state``.
    ``fpn2_of(frame_idx) -> tensor [4096,256] | None`` supplies the level-2 feature of a conditioning frame."""
    st = inference_state
    hdr = {"version": 1, "model": model_name, "layout": "token-major", "num_frames": int(st["num_frames"]),
           "video_height": int(st["video_height"]), "video_width": int(st["video_width"]),
           "obj_ids": [int(o) for o in st["obj_ids"]], "entries": []}
    blobs, off = [], 0

    def add(t):
        nonlocal off
        b = _raw(t)
        rec = {"dtype": _NAME[t.dtype], "shape": list(t.shape), "offset": off, "nbytes": len(b)}
        blobs.append(b)
        pad = (-len(b)) % 64
        if pad:
            blobs.append(b"\0" * pad)
        off += len(b) + pad
        return rec

    for kind, key in (("cond", "cond_frame_outputs"), ("non_cond", "non_cond_frame_outputs")):
        for t, out in st["output_dict"][key].items():
            if out.get("maskmem_features") is None:
                continue
            rec = {"frame": int(t), "kind": kind, "tensors": {k: add(out[k]) for k in ENTRY_TENSORS}}
            if kind == "cond" and fpn2_of is not None:
                f2 = fpn2_of(int(t))
                if f2 is not None:
                    rec["tensors"]["fpn2"] = add(f2)
            hdr["entries"].append(rec)
    js = json.dumps(hdr).encode()
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(js)))
        f.write(js)
        f.write(b"\0" * ((-(16 + len(js))) % 64))
        for b in blobs:
            f.write(b)
    return hdr


def _new_state(num_frames, vh, vw, obj_ids):
    st = {"images": torch.empty((0, 3, 1024, 1024), dtype=torch.float16), "num_frames": int(num_frames), "images_idx": [],
          "offload_video_to_cpu": False, "offload_state_to_cpu": False, "video_height": int(vh), "video_width": int(vw),
          "device": "cpu", "storage_device": "cpu", "point_inputs_per_obj": {}, "mask_inputs_per_obj": {},
          "cached_features": {}, "constants": {}, "obj_id_to_idx": OrderedDict(), "obj_idx_to_id": OrderedDict(),
          "obj_ids": [], "output_dict": {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}},
          "output_dict_per_obj": {}, "temp_output_dict_per_obj": {},
          "consolidated_frame_inds": {"cond_frame_outputs": set(), "non_cond_frame_outputs": set()},
          "tracking_has_started": True, "frames_already_tracked": {}, "preloading_memory_cond_frame_idx": None,
          "preloading_memory_non_cond_frames_idx": None, "max_update_length_for_new_obj_id": 100,
          "preload_fpn2": {}}
    for i, oid in enumerate(obj_ids):
        st["obj_id_to_idx"][oid] = i
        st["obj_idx_to_id"][i] = oid
        st["point_inputs_per_obj"][i] = {}
        st["mask_inputs_per_obj"][i] = {}
        st["output_dict_per_obj"][i] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
        st["temp_output_dict_per_obj"][i] = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
    st["obj_ids"] = list(obj_ids)
    return st


def _read_ds2bank(f) -> dict:
    (n,) = struct.unpack("<Q", f.read(8))
    if n > (64 << 20):
        raise ValueError("DS2BANK header too large")
    hdr = json.loads(f.read(n).decode())
    if hdr.get("version") != 1 or hdr.get("layout") != "token-major":
        raise ValueError(f"unsupported DS2BANK version/layout: {hdr.get('version')}/{hdr.get('layout')}")
    f.seek((16 + n + 63) // 64 * 64)
    payload = f.read()
    st = _new_state(hdr["num_frames"], hdr["video_height"], hdr["video_width"], hdr["obj_ids"])
    st["model"] = hdr.get("model")

    def get(rec):
        dt, es = _DT[rec["dtype"]]
        numel = int(np.prod(rec["shape"])) if rec["shape"] else 1
        if rec["nbytes"] != numel * es or rec["offset"] + rec["nbytes"] > len(payload) or rec["offset"] % 64:
            raise ValueError("corrupt DS2BANK tensor record")
        buf = torch.frombuffer(bytearray(payload[rec["offset"]: rec["offset"] + rec["nbytes"]]), dtype=torch.uint8)
        return buf.view(dt).reshape(rec["shape"])

    for e in hdr["entries"]:
        key = "cond_frame_outputs" if e["kind"] == "cond" else "non_cond_frame_outputs"
        out = {k: get(e["tensors"][k]) for k in ENTRY_TENSORS}
        out["maskmem_pos_enc"] = None
        B = out["obj_ptr"].shape[0]
        if tuple(out["maskmem_features"].shape) != (B, 4096, 64) or tuple(out["pred_masks"].shape) != (B, 1, 256, 256):
            raise ValueError(f"DS2BANK entry of frame {e['frame']} has unexpected shapes")
        st["output_dict"][key][int(e["frame"])] = out
        st["frames_already_tracked"][int(e["frame"])] = {"reverse": True}    # as in a reference pickle (:1013-1016)
        if e["kind"] == "cond":
            st["consolidated_frame_inds"]["cond_frame_outputs"].add(int(e["frame"]))
            if "fpn2" in e["tensors"]:
                st["preload_fpn2"][int(e["frame"])] = get(e["tensors"]["fpn2"])
    return st


def _safe_load_storage(b):
    """Stand-in for ``torch.storage._load_from_bytes`` (which is ``torch.load(..., weights_only=False)``, i.e. a second,
    unrestricted pickle on the file's bytes): the nested stream is read with torch's own weights-only unpickler."""
    return torch.load(io.BytesIO(b), weights_only=True, map_location="cpu")


class _RestrictedUnpickler(pickle.Unpickler):
    """Reference banks are ``pickle.dump(inference_state)``: dicts / OrderedDict / sets / lists of torch tensors,
    ``torch.device`` objects, ints.  Nothing else is allowed to be constructed - including through the nested pickle
    inside a tensor's storage bytes (``_safe_load_storage``)."""
    _OK = {("collections", "OrderedDict"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"),
           ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch", "device"),
           ("torch", "Size"), ("torch.serialization", "_get_layout"),
           ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
           ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}
    _TORCH_TYPES = {"FloatStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "BoolStorage",
                    "ByteStorage", "UntypedStorage", "float32", "float16", "bfloat16", "int64", "int32", "uint8", "bool"}

    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _safe_load_storage
        if (module, name) in self._OK or (module == "torch" and name in self._TORCH_TYPES):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"bank file: refusing to unpickle {module}.{name}")


def _from_reference_pickle(raw: bytes) -> dict:
    """A bank pickled by the reference (or by round 1 of this framework): channel-major ``maskmem_features``
    [B,64,64,64] -> token-major bf16 [B,4096,64]; ``maskmem_pos_enc`` (a model constant) dropped; per-object views
    are rebuilt by ``init_preloading_state``; the frames it carries are kept (they serve the A17 path)."""
    st = _RestrictedUnpickler(io.BytesIO(raw)).load()
    if not isinstance(st, dict) or "output_dict" not in st:
        raise ValueError("not an inference-state pickle")
    for key in ("cond_frame_outputs", "non_cond_frame_outputs"):
        for t, out in st["output_dict"][key].items():
            f = out.get("maskmem_features")
            if f is not None and f.dim() == 4:                     # reference layout [B,C,64,64]
                out["maskmem_features"] = f.flatten(2).transpose(1, 2).contiguous().to(torch.bfloat16)
            elif f is not None and not (f.dim() == 3 and f.shape[1:] == (4096, 64)):
                raise ValueError(f"bank entry of frame {t}: maskmem_features has shape {tuple(f.shape)}; expected "
                                 "[B,64,64,64] (reference) or [B,4096,64] (token-major)")
            out["maskmem_pos_enc"] = None
            for k in ("pred_masks", "obj_ptr", "object_score_logits"):
                if out.get(k) is not None:
                    out[k] = out[k].to(torch.float32)
    for k in ("device", "storage_device"):
        st[k] = "cpu"
    st["cached_features"] = {}
    st["output_dict_per_obj"] = {i: {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}} for i in st["obj_idx_to_id"]}
    st.setdefault("preload_fpn2", {})
    return st


def load_bank(path, allow_pickle=None) -> dict:
    """-> host-resident inference state (DS2BANK or reference pickle); ``init_preloading_state`` moves it to the GPU.
    ``allow_pickle=False`` (or ``DS2_BANK_ALLOW_PICKLE=0``) refuses anything that is not a DS2BANK file; by default a
    file without the magic is read as a reference bank through the restricted unpickler."""
    if allow_pickle is None:
        allow_pickle = os.environ.get("DS2_BANK_ALLOW_PICKLE", "1") not in ("", "0")
    with open(path, "rb") as f:
        head = f.read(8)
        if head == MAGIC:
            return _read_ds2bank(f)
        if not allow_pickle:
            raise ValueError(f"{path}: not a DS2BANK file and reference pickles are disabled (allow_pickle=False)")
        raw = head + f.read()
    return _from_reference_pickle(raw)
