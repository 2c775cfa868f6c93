// PyTorch-ROCm custom ops over the C-ABI: TORCH_LIBRARY(det_sam2, m).
//
// The north-star asks for the HIP stages to be "called from Python via PyTorch-ROCm custom ops"; the reference's own
// native precedent is a torch extension (sam2/csrc/connected_components.cu:284-289, pybind `get_connected_componnets`).
// This file is host-only C++ (no device code): every op checks its at::Tensor arguments (device, dtype, shape,
// contiguity), allocates its outputs with ATen on the inputs' device, takes the CURRENT HIP stream of that device from
// c10 and forwards to the extern "C" entry point of include/detsam2_hip.h - the C-ABI stays the one stable boundary, the
// ops add dispatcher visibility (torch.ops.det_sam2.*), tensor-lifetime safety (no raw data_ptr in Python) and stream
// correctness by construction.  A ds2_model travels as an int64 handle (the pointer ds2_model_create returned).
// Built into det-sam2_amd/lib/libdetsam2_torch.so by __graft_entry__.build(); loaded with torch.ops.load_library.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // PyTorch-ROCm tensors carry DeviceType::CUDA ("HIP masquerading as
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>      // CUDA"): guard / current stream come from the masquerading forms
#include <torch/library.h>

#include <cmath>
#include <tuple>
#include <vector>

#include "../../include/detsam2_hip.h"

namespace {

using at::Tensor;

ds2_model* model_of(int64_t h) {
  TORCH_CHECK(h != 0, "det_sam2: null model handle");
  return reinterpret_cast<ds2_model*>(static_cast<intptr_t>(h));
}
void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(); }
void check(int rc, const char* what) {
  TORCH_CHECK(rc == DS2_OK, "det_sam2::", what, " failed (code ", rc, "): ", ds2_last_error());   // c10::Error -> RuntimeError
}
void want(const Tensor& t, at::ScalarType dt, const char* name) {
  TORCH_CHECK(t.is_cuda(), "det_sam2: ", name, " must be a GPU tensor");
  TORCH_CHECK(t.scalar_type() == dt, "det_sam2: ", name, " must be ", toString(dt), ", got ", toString(t.scalar_type()));
  TORCH_CHECK(t.is_contiguous(), "det_sam2: ", name, " must be contiguous");
}
const float* fptr(const c10::optional<Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }

constexpr int64_t TOK = 4096;

// A3: load_video_frames (sam2/utils/misc.py:280-284,328-342,358-359)
Tensor ingest_frames(int64_t model, const Tensor& frames_u8) {
  want(frames_u8, at::kByte, "frames_u8");
  TORCH_CHECK(frames_u8.dim() == 4 && frames_u8.size(3) == 3, "det_sam2::ingest_frames: frames_u8 must be [n,H,W,3]");
  c10::hip::HIPGuardMasqueradingAsCUDA g(frames_u8.device());
  const int64_t n = frames_u8.size(0);
  Tensor out = at::empty({n, 3, 1024, 1024}, frames_u8.options().dtype(at::kHalf));
  check(ds2_ingest_frames(model_of(model), frames_u8.data_ptr<uint8_t>(), (int32_t)n, (int32_t)frames_u8.size(1),
                          (int32_t)frames_u8.size(2), reinterpret_cast<uint16_t*>(out.data_ptr()), stream_of(frames_u8)),
        "ingest_frames");
  return out;
}

// A4 + A5: SAM2Base.forward_image (sam2_base.py:450-461), n frames per launch
std::tuple<Tensor, Tensor, Tensor> image_encoder(int64_t model, const Tensor& frames) {
  TORCH_CHECK(frames.dim() == 4 && frames.size(1) == 3, "det_sam2::image_encoder: frames must be [n,3,S,S]");
  TORCH_CHECK(frames.scalar_type() == at::kHalf || frames.scalar_type() == at::kFloat, "det_sam2::image_encoder: fp16 or fp32 frames");
  want(frames, frames.scalar_type(), "frames");
  c10::hip::HIPGuardMasqueradingAsCUDA g(frames.device());
  const int64_t n = frames.size(0);
  auto o = frames.options().dtype(at::kFloat);
  Tensor f0 = at::empty({n, 65536, 32}, o), f1 = at::empty({n, 16384, 64}, o), f2 = at::empty({n, TOK, 256}, o);
  if (frames.scalar_type() == at::kHalf)
    check(ds2_image_encoder_batch(model_of(model), reinterpret_cast<const uint16_t*>(frames.data_ptr()), (int32_t)n,
                                  f0.data_ptr<float>(), f1.data_ptr<float>(), f2.data_ptr<float>(), stream_of(frames)), "image_encoder");
  else
    check(ds2_image_encoder_f32(model_of(model), frames.data_ptr<float>(), (int32_t)n, f0.data_ptr<float>(), f1.data_ptr<float>(),
                                f2.data_ptr<float>(), stream_of(frames)), "image_encoder");
  return {f0, f1, f2};
}

// the entry tables of a bank as the C-ABI takes them (host arrays of device pointers)
struct BankTables {
  std::vector<const void*> fp;
  std::vector<int32_t> rows;
  std::vector<const float*> pp;
  std::vector<float> pos;
};
static BankTables bank_tables(int64_t B, at::TensorList feats, at::IntArrayRef tpos_rows, at::TensorList ptrs, at::ArrayRef<double> ptr_pos,
                              const char* who) {
  TORCH_CHECK(feats.size() == tpos_rows.size() && ptrs.size() == ptr_pos.size(), "det_sam2::", who, ": table lengths differ");
  TORCH_CHECK(feats.size() + ptrs.size() > 0, "det_sam2::", who, ": empty bank");
  BankTables t;
  for (size_t i = 0; i < feats.size(); ++i) {
    want(feats[i], at::kBFloat16, "feats[i]");
    TORCH_CHECK(feats[i].dim() == 3 && feats[i].size(0) == B && feats[i].size(1) == TOK && feats[i].size(2) == 64,
                "det_sam2::", who, ": feats[i] must be bf16 [B,4096,64]");
    t.fp.push_back(feats[i].data_ptr());
    t.rows.push_back((int32_t)tpos_rows[i]);
  }
  for (size_t i = 0; i < ptrs.size(); ++i) {
    want(ptrs[i], at::kFloat, "ptrs[i]");
    TORCH_CHECK(ptrs[i].dim() == 2 && ptrs[i].size(0) == B && ptrs[i].size(1) == 256, "det_sam2::", who, ": ptrs[i] must be fp32 [B,256]");
    t.pp.push_back(ptrs[i].data_ptr<float>());
    t.pos.push_back((float)ptr_pos[i]);
  }
  return t;
}

// A11: tensor part of _prepare_memory_conditioned_features (sam2_base.py:565-648)
std::tuple<Tensor, Tensor> bank_assemble(int64_t model, int64_t B, at::TensorList feats, at::IntArrayRef tpos_rows,
                                         at::TensorList ptrs, at::ArrayRef<double> ptr_pos) {
  BankTables t = bank_tables(B, feats, tpos_rows, ptrs, ptr_pos, "bank_assemble");
  const Tensor& any = feats.size() ? feats[0] : ptrs[0];
  c10::hip::HIPGuardMasqueradingAsCUDA g(any.device());
  const int64_t nk = (int64_t)feats.size() * TOK + 4 * (int64_t)ptrs.size();
  auto o = any.options().dtype(at::kFloat);
  Tensor mem = at::empty({B, nk, 64}, o), mpos = at::empty({B, nk, 64}, o);
  check(ds2_bank_assemble(model_of(model), (int32_t)B, (int32_t)t.fp.size(), t.fp.data(), t.rows.data(), (int32_t)t.pp.size(), t.pp.data(),
                          t.pos.data(), mem.data_ptr<float>(), mpos.data_ptr<float>(), stream_of(any)), "bank_assemble");
  return {mem, mpos};
}

// A11 + A12 in one call (the tracking loop): the bank's entries -> memory-conditioned features [B,4096,256]; curr [4096,256] shared
Tensor bank_memory_attention(int64_t model, int64_t B, const Tensor& curr, at::TensorList feats, at::IntArrayRef tpos_rows,
                             at::TensorList ptrs, at::ArrayRef<double> ptr_pos) {
  want(curr, at::kFloat, "curr");
  TORCH_CHECK(curr.dim() == 2 && curr.size(0) == TOK && curr.size(1) == 256, "det_sam2::bank_memory_attention: curr must be [4096,256]");
  BankTables t = bank_tables(B, feats, tpos_rows, ptrs, ptr_pos, "bank_memory_attention");
  c10::hip::HIPGuardMasqueradingAsCUDA g(curr.device());
  Tensor out = at::empty({B, TOK, 256}, curr.options());
  check(ds2_bank_memory_attention(model_of(model), (int32_t)B, curr.data_ptr<float>(), (int32_t)t.fp.size(), t.fp.data(), t.rows.data(),
                                  (int32_t)t.pp.size(), t.pp.data(), t.pos.data(), out.data_ptr<float>(), stream_of(curr)),
        "bank_memory_attention");
  return out;
}

// A12: MemoryAttention.forward (memory_attention.py:119-176).  curr [4096,256] shared by the B objects or [B,4096,256];
// curr_pos None = the model's constant position encoding, else [4096,256] / [B,4096,256].
Tensor memory_attention(int64_t model, int64_t B, const Tensor& curr, const c10::optional<Tensor>& curr_pos, const Tensor& memory,
                        const Tensor& memory_pos, int64_t num_obj_ptr_tokens) {
  want(curr, at::kFloat, "curr");
  want(memory, at::kFloat, "memory");
  want(memory_pos, at::kFloat, "memory_pos");
  TORCH_CHECK(memory.dim() == 3 && memory.size(0) == B && memory.size(2) == 64 && memory_pos.sizes() == memory.sizes(),
              "det_sam2::memory_attention: memory / memory_pos must be fp32 [B,Nk,64]");
  const bool curr_shared = curr.dim() == 2;
  TORCH_CHECK((curr_shared && curr.size(0) == TOK && curr.size(1) == 256) ||
              (curr.dim() == 3 && curr.size(0) == B && curr.size(1) == TOK && curr.size(2) == 256),
              "det_sam2::memory_attention: curr must be [4096,256] or [B,4096,256]");
  bool pos_shared = true;
  if (curr_pos.has_value()) {
    want(*curr_pos, at::kFloat, "curr_pos");
    pos_shared = curr_pos->dim() == 2;
    TORCH_CHECK(curr_pos->numel() == (pos_shared ? 1 : B) * TOK * 256, "det_sam2::memory_attention: bad curr_pos shape");
  }
  c10::hip::HIPGuardMasqueradingAsCUDA g(curr.device());
  Tensor out = at::empty({B, TOK, 256}, curr.options());
  check(ds2_memory_attention_ex(model_of(model), (int32_t)B, curr.data_ptr<float>(), curr_shared ? 1 : 0, fptr(curr_pos),
                                pos_shared ? 1 : 0, memory.data_ptr<float>(), memory_pos.data_ptr<float>(), (int32_t)memory.size(1),
                                (int32_t)num_obj_ptr_tokens, out.data_ptr<float>(), stream_of(curr)), "memory_attention");
  return out;
}

// A7 + A8: SAM2Base._forward_sam_heads (sam2_base.py:254-397) -> (low_res [B,256,256], obj_ptr [B,256], obj_logits [B], ious [B])
std::tuple<Tensor, Tensor, Tensor, Tensor> sam_heads(int64_t model, int64_t B, const Tensor& pix_feat, bool pix_bcast, bool add_no_mem_embed,
                                                     const Tensor& fpn0, const Tensor& fpn1, const c10::optional<Tensor>& point_coords,
                                                     const c10::optional<Tensor>& point_labels, const c10::optional<Tensor>& mask_inputs,
                                                     bool multimask) {
  want(pix_feat, at::kFloat, "pix_feat");
  want(fpn0, at::kFloat, "fpn0");
  want(fpn1, at::kFloat, "fpn1");
  TORCH_CHECK(pix_feat.numel() == (pix_bcast ? 1 : B) * TOK * 256, "det_sam2::sam_heads: pix_feat must be [B,4096,256] ([4096,256] with pix_bcast)");
  TORCH_CHECK(fpn0.numel() == 65536 * 32 && fpn1.numel() == 16384 * 64, "det_sam2::sam_heads: fpn0 [65536,32] / fpn1 [16384,64] expected");
  TORCH_CHECK(point_coords.has_value() == point_labels.has_value(), "det_sam2::sam_heads: point_coords and point_labels go together");
  int64_t P = 0;
  const int32_t* labels = nullptr;
  if (point_coords.has_value()) {
    want(*point_coords, at::kFloat, "point_coords");
    want(*point_labels, at::kInt, "point_labels");
    TORCH_CHECK(point_coords->dim() == 3 && point_coords->size(0) == B && point_coords->size(2) == 2 && point_labels->dim() == 2 &&
                point_labels->size(0) == B && point_labels->size(1) == point_coords->size(1),
                "det_sam2::sam_heads: point_coords [B,P,2] / point_labels [B,P] expected");
    P = point_coords->size(1);
    labels = point_labels->data_ptr<int32_t>();
  }
  if (mask_inputs.has_value()) {
    want(*mask_inputs, at::kFloat, "mask_inputs");
    TORCH_CHECK(mask_inputs->numel() == B * 256 * 256, "det_sam2::sam_heads: mask_inputs must be [B,256,256]");
  }
  c10::hip::HIPGuardMasqueradingAsCUDA g(pix_feat.device());
  auto o = pix_feat.options();
  Tensor low = at::empty({B, 256, 256}, o), ptr = at::empty({B, 256}, o), obj = at::empty({B}, o), iou = at::empty({B}, o);
  check(ds2_sam_heads_mask(model_of(model), (int32_t)B, pix_feat.data_ptr<float>(), pix_bcast ? 1 : 0, add_no_mem_embed ? 1 : 0,
                           fpn0.data_ptr<float>(), fpn1.data_ptr<float>(), P ? fptr(point_coords) : nullptr, P ? labels : nullptr, (int32_t)P,
                           fptr(mask_inputs), multimask ? 1 : 0, low.data_ptr<float>(), ptr.data_ptr<float>(), obj.data_ptr<float>(),
                           iou.data_ptr<float>(), stream_of(pix_feat)), "sam_heads");
  return {low, ptr, obj, iou};
}

// A7: PromptEncoder.forward (prompt_encoder.py:134-171) -> (sparse [B,Ns,256], dense [B,4096,256] token-major)
std::tuple<Tensor, Tensor> prompt_encoder(int64_t model, int64_t B, const c10::optional<Tensor>& point_coords,
                                          const c10::optional<Tensor>& point_labels, bool pad, const c10::optional<Tensor>& mask_inputs,
                                          const Tensor& like) {
  TORCH_CHECK(like.is_cuda() && B > 0, "det_sam2::prompt_encoder: `like` must be a device tensor");
  TORCH_CHECK(point_coords.has_value() == point_labels.has_value(), "det_sam2::prompt_encoder: point_coords and point_labels go together");
  int64_t P = 0;
  const int32_t* labels = nullptr;
  if (point_coords.has_value()) {
    want(*point_coords, at::kFloat, "point_coords");
    want(*point_labels, at::kInt, "point_labels");
    TORCH_CHECK(point_coords->dim() == 3 && point_coords->size(0) == B && point_coords->size(2) == 2 && point_labels->dim() == 2 &&
                point_labels->size(0) == B && point_labels->size(1) == point_coords->size(1),
                "det_sam2::prompt_encoder: point_coords [B,P,2] / point_labels [B,P] expected");
    P = point_coords->size(1);
    labels = point_labels->data_ptr<int32_t>();
  }
  if (mask_inputs.has_value()) {
    want(*mask_inputs, at::kFloat, "mask_inputs");
    TORCH_CHECK(mask_inputs->numel() == B * 256 * 256, "det_sam2::prompt_encoder: mask_inputs must be [B,256,256]");
  }
  c10::hip::HIPGuardMasqueradingAsCUDA g(like.device());
  auto o = like.options().dtype(at::kFloat);
  const int64_t Ns = P ? P + (pad ? 1 : 0) : 0;
  Tensor sparse = at::empty({B, Ns, 256}, o), dense = at::empty({B, TOK, 256}, o);
  check(ds2_prompt_encoder(model_of(model), (int32_t)B, P ? fptr(point_coords) : nullptr, P ? labels : nullptr, (int32_t)P, pad ? 1 : 0,
                           fptr(mask_inputs), Ns ? sparse.data_ptr<float>() : nullptr, dense.data_ptr<float>(), stream_of(like)),
        "prompt_encoder");
  return {sparse, dense};
}

// A8: MaskDecoder.predict_masks (mask_decoder.py:163-259) -> (masks [B,4,256,256], iou [B,4], mask_tokens [B,4,256], obj [B])
std::tuple<Tensor, Tensor, Tensor, Tensor> mask_decoder(int64_t model, int64_t B, const Tensor& image_embeddings, const Tensor& image_pe,
                                                        const Tensor& sparse, const Tensor& dense, const Tensor& feat_s0,
                                                        const Tensor& feat_s1) {
  want(image_embeddings, at::kFloat, "image_embeddings");
  want(image_pe, at::kFloat, "image_pe");
  want(sparse, at::kFloat, "sparse");
  want(dense, at::kFloat, "dense");
  want(feat_s0, at::kFloat, "feat_s0");
  want(feat_s1, at::kFloat, "feat_s1");
  TORCH_CHECK(image_embeddings.numel() == B * TOK * 256 && dense.numel() == B * TOK * 256 && image_pe.numel() == TOK * 256,
              "det_sam2::mask_decoder: image_embeddings / dense [B,4096,256], image_pe [4096,256] expected");
  TORCH_CHECK(sparse.dim() == 3 && sparse.size(0) == B && sparse.size(2) == 256, "det_sam2::mask_decoder: sparse [B,Ns,256] expected");
  TORCH_CHECK(feat_s0.numel() == 65536 * 32 && feat_s1.numel() == 16384 * 64, "det_sam2::mask_decoder: feat_s0 [65536,32] / feat_s1 [16384,64] expected");
  c10::hip::HIPGuardMasqueradingAsCUDA g(image_embeddings.device());
  auto o = image_embeddings.options();
  const int64_t Ns = sparse.size(1);
  Tensor masks = at::empty({B, 4, 256, 256}, o), iou = at::empty({B, 4}, o), tok = at::empty({B, 4, 256}, o), obj = at::empty({B}, o);
  check(ds2_mask_decoder(model_of(model), (int32_t)B, image_embeddings.data_ptr<float>(), image_pe.data_ptr<float>(),
                         Ns ? sparse.data_ptr<float>() : nullptr, (int32_t)Ns, dense.data_ptr<float>(), feat_s0.data_ptr<float>(),
                         feat_s1.data_ptr<float>(), masks.data_ptr<float>(), iou.data_ptr<float>(), tok.data_ptr<float>(),
                         obj.data_ptr<float>(), stream_of(image_embeddings)), "mask_decoder");
  return {masks, iou, tok, obj};
}

// A13: SAM2Base._encode_new_memory (sam2_base.py:692-743) -> maskmem bf16 [B,4096,64]
Tensor memory_encoder(int64_t model, int64_t B, const Tensor& fpn2, const Tensor& low_res, const Tensor& obj_logits, bool binarize) {
  want(fpn2, at::kFloat, "fpn2");
  want(low_res, at::kFloat, "low_res");
  want(obj_logits, at::kFloat, "obj_logits");
  TORCH_CHECK(fpn2.numel() == TOK * 256 && low_res.numel() == B * 65536 && obj_logits.numel() == B, "det_sam2::memory_encoder: bad shapes");
  c10::hip::HIPGuardMasqueradingAsCUDA g(fpn2.device());
  Tensor out = at::empty({B, TOK, 64}, fpn2.options().dtype(at::kBFloat16));
  check(ds2_memory_encoder(model_of(model), (int32_t)B, fpn2.data_ptr<float>(), low_res.data_ptr<float>(), obj_logits.data_ptr<float>(),
                           binarize ? 1 : 0, reinterpret_cast<uint16_t*>(out.data_ptr()), stream_of(fpn2)), "memory_encoder");
  return out;
}

// MemoryEncoder.forward itself (memory_encoder.py:158-181): pix_feat [B,4096,256] (or shared [4096,256]), masks fp32
// [B,1024,1024] -> vision_features fp32 [B,4096,64] (no no_obj_embed_spatial, no bf16 rounding)
Tensor memory_encoder_module(int64_t model, int64_t B, const Tensor& pix_feat, const Tensor& masks, bool skip_mask_sigmoid) {
  want(pix_feat, at::kFloat, "pix_feat");
  want(masks, at::kFloat, "masks");
  const bool shared = pix_feat.numel() == TOK * 256 && B != 1;
  TORCH_CHECK(pix_feat.numel() == (shared ? 1 : B) * TOK * 256 && masks.numel() == B * 1024 * 1024, "det_sam2::memory_encoder_module: bad shapes");
  c10::hip::HIPGuardMasqueradingAsCUDA g(pix_feat.device());
  Tensor out = at::empty({B, TOK, 64}, pix_feat.options());
  check(ds2_memory_encoder_ex(model_of(model), (int32_t)B, pix_feat.data_ptr<float>(), shared ? 1 : 0, masks.data_ptr<float>(),
                              skip_mask_sigmoid ? 1 : 0, out.data_ptr<float>(), stream_of(pix_feat)), "memory_encoder_module");
  return out;
}

// F3: F.interpolate(bilinear, antialias=True, align_corners=False) of (x*in_scale + in_bias) [B,Hin,Win] -> [B,Hout,Wout];
// threshold < inf binarises (sam2_video_predictor.py:552-561, sam2_base.py:407-415)
Tensor resize_aa(const Tensor& x, int64_t Hout, int64_t Wout, double in_scale, double in_bias, double threshold) {
  want(x, at::kFloat, "x");
  TORCH_CHECK(x.dim() == 3, "det_sam2::resize_aa: x must be [B,Hin,Win]");
  c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
  const int64_t B = x.size(0), Hin = x.size(1), Win = x.size(2);
  Tensor work = at::empty({B, Hin, Wout}, x.options()), out = at::empty({B, Hout, Wout}, x.options());
  check(ds2_resize_aa(x.data_ptr<float>(), (int32_t)B, (int32_t)Hin, (int32_t)Win, (int32_t)Hout, (int32_t)Wout, (float)in_scale,
                      (float)in_bias, std::isinf(threshold) ? INFINITY : (float)threshold, work.data_ptr<float>(), out.data_ptr<float>(),
                      stream_of(x)), "resize_aa");
  return out;
}

// F3: mask_downsample conv + any(mask > 0) (sam2_base.py:425-429,436-440) -> (mask_ds [B,256,256], obj_logits [B])
std::tuple<Tensor, Tensor> mask_prompt_prepare(int64_t model, const Tensor& mask) {
  want(mask, at::kFloat, "mask");
  TORCH_CHECK(mask.dim() == 3 && mask.size(1) == 1024 && mask.size(2) == 1024, "det_sam2::mask_prompt_prepare: mask must be [B,1024,1024]");
  c10::hip::HIPGuardMasqueradingAsCUDA g(mask.device());
  const int64_t B = mask.size(0);
  Tensor ds = at::empty({B, 256, 256}, mask.options()), obj = at::empty({B}, mask.options()), work = at::empty({B}, mask.options().dtype(at::kInt));
  check(ds2_mask_prompt_prepare(model_of(model), (int32_t)B, mask.data_ptr<float>(), ds.data_ptr<float>(), obj.data_ptr<float>(),
                                work.data_ptr<int32_t>(), stream_of(mask)), "mask_prompt_prepare");
  return {ds, obj};
}

// F3: obj_ptr = lam*obj_ptr + (1-lam)*no_obj_ptr, lam = obj_logits > 0 (sam2_base.py:441-444); functional
Tensor obj_ptr_gate(int64_t model, const Tensor& obj_ptr, const Tensor& obj_logits) {
  want(obj_ptr, at::kFloat, "obj_ptr");
  want(obj_logits, at::kFloat, "obj_logits");
  TORCH_CHECK(obj_ptr.dim() == 2 && obj_ptr.size(1) == 256 && obj_logits.numel() == obj_ptr.size(0), "det_sam2::obj_ptr_gate: bad shapes");
  c10::hip::HIPGuardMasqueradingAsCUDA g(obj_ptr.device());
  Tensor out = obj_ptr.clone();
  check(ds2_obj_ptr_gate(model_of(model), (int32_t)out.size(0), out.data_ptr<float>(), obj_logits.data_ptr<float>(), stream_of(out)), "obj_ptr_gate");
  return out;
}

// A15: _get_orig_video_res_output + `> 0` + bit-pack (sam2_video_predictor.py:618-642, det_sam2_RT.py:396-399)
std::tuple<Tensor, Tensor> mask_output(int64_t model, const Tensor& low_res, int64_t Hv, int64_t Wv, bool want_logits, bool want_packed) {
  want(low_res, at::kFloat, "low_res");
  TORCH_CHECK(low_res.dim() == 3 && low_res.size(1) == 256 && low_res.size(2) == 256, "det_sam2::mask_output: low_res must be [B,256,256]");
  TORCH_CHECK(want_logits || want_packed, "det_sam2::mask_output: nothing requested");
  c10::hip::HIPGuardMasqueradingAsCUDA g(low_res.device());
  const int64_t B = low_res.size(0);
  Tensor logits = want_logits ? at::empty({B, 1, Hv, Wv}, low_res.options()) : at::empty({0}, low_res.options());
  Tensor packed = want_packed ? at::empty({B, Hv, (Wv + 7) / 8}, low_res.options().dtype(at::kByte)) : at::empty({0}, low_res.options().dtype(at::kByte));
  check(ds2_mask_output(model_of(model), low_res.data_ptr<float>(), (int32_t)B, (int32_t)Hv, (int32_t)Wv,
                        want_logits ? logits.data_ptr<float>() : nullptr, want_packed ? packed.data_ptr<uint8_t>() : nullptr,
                        stream_of(low_res)), "mask_output");
  return {logits, packed};
}

// A14: the reference's one native op, same name (and spelling) and contract: uint8 [N,1,H,W] -> [labels int32, counts int32]
// (sam2/csrc/connected_components.cu:213-289); any H, W here.
std::vector<Tensor> get_connected_componnets(const Tensor& inputs) {
  TORCH_CHECK(inputs.is_cuda(), "inputs must be a CUDA tensor");                  // the reference's AT_ASSERTM messages (:215-227)
  TORCH_CHECK(inputs.dim() == 4, "inputs must be [N, 1, H, W] shape");
  TORCH_CHECK(inputs.scalar_type() == at::kByte, "inputs must be a uint8 type");
  TORCH_CHECK(inputs.size(1) == 1, "inputs must be [N, 1, H, W] shape");
  c10::hip::HIPGuardMasqueradingAsCUDA g(inputs.device());
  Tensor in = inputs.contiguous();
  const int64_t N = in.size(0), H = in.size(2), W = in.size(3);
  auto o = in.options().dtype(at::kInt);
  Tensor labels = at::empty({N, 1, H, W}, o), counts = at::empty({N, 1, H, W}, o), work = at::empty({2 * N * H * W}, o);
  check(ds2_connected_components(in.data_ptr<uint8_t>(), (int32_t)N, (int32_t)H, (int32_t)W, labels.data_ptr<int32_t>(),
                                 counts.data_ptr<int32_t>(), work.data_ptr<int32_t>(), stream_of(in)), "get_connected_componnets");
  return {labels, counts};
}

// A14: fill_holes_in_mask_scores (sam2/utils/misc.py:365-393), functional form
Tensor fill_holes(const Tensor& logits, int64_t max_area) {
  want(logits, at::kFloat, "logits");
  TORCH_CHECK(logits.dim() >= 2, "det_sam2::fill_holes: logits must be [..., H, W]");
  c10::hip::HIPGuardMasqueradingAsCUDA g(logits.device());
  Tensor out = logits.clone();
  const int64_t H = out.size(-2), W = out.size(-1), N = out.numel() / (H * W);
  Tensor work = at::empty({3 * N * H * W}, out.options().dtype(at::kInt));
  check(ds2_fill_holes(out.data_ptr<float>(), (int32_t)N, (int32_t)H, (int32_t)W, (int32_t)max_area, work.data_ptr<int32_t>(), stream_of(out)),
        "fill_holes");
  return out;
}

// F4 detector half: ultralytics ops.non_max_suppression + scale_boxes / clip_boxes on the raw YOLOv8 head output
// pred fp32 [nb, 4+nc, N] -> (dets fp32 [nb, max_det, 6] = xyxy, conf, cls; counts int32 [nb], -1 = candidate overflow)
std::tuple<Tensor, Tensor> yolo_postprocess(const Tensor& pred, double conf_thres, double iou_thres, int64_t max_det,
                                            const c10::optional<Tensor>& scale5) {
  want(pred, at::kFloat, "pred");
  TORCH_CHECK(pred.dim() == 3 && pred.size(1) > 4, "det_sam2::yolo_postprocess: pred must be [nb, 4+nc, N]");
  if (scale5.has_value()) {
    want(*scale5, at::kFloat, "scale5");
    TORCH_CHECK(scale5->numel() == 5, "det_sam2::yolo_postprocess: scale5 = gain, pad_x, pad_y, orig_w, orig_h");
  }
  c10::hip::HIPGuardMasqueradingAsCUDA g(pred.device());
  const int64_t nb = pred.size(0), nc = pred.size(1) - 4, N = pred.size(2);
  const int64_t wb = ds2_yolo_postprocess_work_bytes((int32_t)nb, (int32_t)N);
  Tensor work = at::empty({wb}, pred.options().dtype(at::kByte));
  Tensor dets = at::zeros({nb, max_det, 6}, pred.options()), counts = at::empty({nb}, pred.options().dtype(at::kInt));
  check(ds2_yolo_postprocess(pred.data_ptr<float>(), (int32_t)nb, (int32_t)nc, (int32_t)N, (float)conf_thres, (float)iou_thres,
                             (int32_t)max_det, fptr(scale5), dets.data_ptr<float>(), counts.data_ptr<int32_t>(), work.data_ptr(), wb,
                             stream_of(pred)), "yolo_postprocess");
  return {dets, counts};
}

}  // namespace

TORCH_LIBRARY(det_sam2, m) {
  m.def("ingest_frames(int model, Tensor frames_u8) -> Tensor");
  m.def("image_encoder(int model, Tensor frames) -> (Tensor, Tensor, Tensor)");
  m.def("bank_assemble(int model, int B, Tensor[] feats, int[] tpos_rows, Tensor[] ptrs, float[] ptr_pos) -> (Tensor, Tensor)");
  m.def("bank_memory_attention(int model, int B, Tensor curr, Tensor[] feats, int[] tpos_rows, Tensor[] ptrs, float[] ptr_pos) -> Tensor");
  m.def("memory_attention(int model, int B, Tensor curr, Tensor? curr_pos, Tensor memory, Tensor memory_pos, int num_obj_ptr_tokens) -> Tensor");
  m.def("sam_heads(int model, int B, Tensor pix_feat, bool pix_bcast, bool add_no_mem_embed, Tensor fpn0, Tensor fpn1, "
        "Tensor? point_coords, Tensor? point_labels, Tensor? mask_inputs, bool multimask) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("prompt_encoder(int model, int B, Tensor? point_coords, Tensor? point_labels, bool pad, Tensor? mask_inputs, Tensor like) -> (Tensor, Tensor)");
  m.def("mask_decoder(int model, int B, Tensor image_embeddings, Tensor image_pe, Tensor sparse, Tensor dense, Tensor feat_s0, Tensor feat_s1) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("memory_encoder(int model, int B, Tensor fpn2, Tensor low_res, Tensor obj_logits, bool binarize) -> Tensor");
  m.def("memory_encoder_module(int model, int B, Tensor pix_feat, Tensor masks, bool skip_mask_sigmoid) -> Tensor");
  m.def("resize_aa(Tensor x, int Hout, int Wout, float in_scale, float in_bias, float threshold) -> Tensor");
  m.def("mask_prompt_prepare(int model, Tensor mask) -> (Tensor, Tensor)");
  m.def("obj_ptr_gate(int model, Tensor obj_ptr, Tensor obj_logits) -> Tensor");
  m.def("mask_output(int model, Tensor low_res, int Hv, int Wv, bool want_logits, bool want_packed) -> (Tensor, Tensor)");
  m.def("get_connected_componnets(Tensor inputs) -> Tensor[]");
  m.def("fill_holes(Tensor logits, int max_area) -> Tensor");
  m.def("yolo_postprocess(Tensor pred, float conf_thres, float iou_thres, int max_det, Tensor? scale5) -> (Tensor, Tensor)");
}

// GPU tensors on PyTorch-ROCm dispatch under the CUDA key (HIP masquerades as CUDA); a CPU tensor finds no kernel and the
// dispatcher raises NotImplementedError - there is no CPU path.
TORCH_LIBRARY_IMPL(det_sam2, CUDA, m) {
  m.impl("ingest_frames", &ingest_frames);
  m.impl("image_encoder", &image_encoder);
  m.impl("bank_assemble", &bank_assemble);
  m.impl("bank_memory_attention", &bank_memory_attention);
  m.impl("memory_attention", &memory_attention);
  m.impl("sam_heads", &sam_heads);
  m.impl("prompt_encoder", &prompt_encoder);
  m.impl("mask_decoder", &mask_decoder);
  m.impl("memory_encoder", &memory_encoder);
  m.impl("memory_encoder_module", &memory_encoder_module);
  m.impl("resize_aa", &resize_aa);
  m.impl("mask_prompt_prepare", &mask_prompt_prepare);
  m.impl("obj_ptr_gate", &obj_ptr_gate);
  m.impl("mask_output", &mask_output);
  m.impl("get_connected_componnets", &get_connected_componnets);
  m.impl("fill_holes", &fill_holes);
  m.impl("yolo_postprocess", &yolo_postprocess);
}
