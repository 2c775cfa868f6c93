/* detsam2_hip.h - C-ABI of libdetsam2_hip.so: the MI355X (gfx950) implementation of the Det-SAM2
 * per-frame hot path (SAM 2.1 video predictor).  Plain pointers and sizes only; every pointer is a
 * DEVICE pointer unless stated otherwise; `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  All entry points are asynchronous on `stream` and return 0 on success; on failure they
 * return a non-zero code and ds2_last_error() describes it.
 *
 * The reference has no FFI of its own for this path (its only native symbol is the CUDA
 * connected-components op, sam2/csrc/connected_components.cu:284-289); its plugin seam is the
 * Hydra `_target_` module tree (sam2/configs/sam2.1/sam2.1_hiera_l.yaml:5-80).  Each stage below
 * therefore replaces one nn.Module.forward / SAM2Base method of that tree, cited per function;
 * file:line are relative to the reference repository.  INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Layout conventions: activations are fp32 and TOKEN-MAJOR ("NHWC"): a feature map
 * that the reference holds as [B,C,H,W] is [B, H*W, C] here; memory-bank features are bf16
 * [B, 4096, 64] (the reference stores bf16 [B,64,64,64], sam2_video_predictor.py:1337).
 */
#ifndef DETSAM2_HIP_H
#define DETSAM2_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS2_ABI_VERSION 1

/* return codes of every int-returning entry point (0 = success; ds2_last_error() holds the message otherwise) */
#ifndef DS2_OK
#define DS2_OK 0
#define DS2_ERR_ARG 1          /* bad argument */
#define DS2_ERR_HIP 2          /* a HIP runtime call or kernel launch failed */
#define DS2_ERR_STATE 3        /* model not finalized / missing parameter / workspace exhausted */
#define DS2_ERR_UNSUPPORTED 4  /* no kernel for the requested shape */
#endif

typedef struct ds2_model ds2_model;

/* Hyper-parameters (sam2/configs/sam2.1/sam2.1_hiera_*.yaml + build_sam.py:126-135). */
typedef struct ds2_config {
  int32_t image_size;             /* 1024 */
  int32_t embed_dim, num_heads;   /* Hiera trunk (hieradet.py:172-196) */
  int32_t stages[4];
  int32_t global_att_blocks[4];
  int32_t n_global_att_blocks;
  int32_t window_spec[4];
  int32_t d_model, mem_dim;       /* 256, 64 */
  int32_t num_maskmem;            /* 7 */
  int32_t mem_attn_layers;        /* 4 */
  int32_t mem_attn_ffn;           /* 2048 */
  int32_t max_batch;              /* workspace is sized for this many objects (it grows on demand) */
  float sigmoid_scale_for_mem_enc, sigmoid_bias_for_mem_enc;            /* 20, -10 */
  float dynamic_multimask_stability_delta, dynamic_multimask_stability_thresh; /* 0.05, 0.98 */
} ds2_config;

const char* ds2_last_error(void);   /* thread-local, valid until the next failing call on this thread */
int ds2_abi_version(void);

/* ---- model lifetime ------------------------------------------------------------------------
 * Replaces build_sam2_video_predictor + _load_checkpoint (sam2/build_sam.py:111-178): parameters are
 * registered under their state_dict key (fp32, host or device pointer; the library copies them), then
 * ds2_model_finalize checks the set is complete (strict, like load_state_dict) and packs derived
 * weights.  Names starting with '#' are host-precomputed constants of the model (see
 * det-sam2_amd/constants.py): "#pos_embed" [65536,embed_dim], "#rope_cis" [4096,128,2],
 * "#vision_pos" [4096,256], "#maskmem_pos" [4096,64], "#dense_pe" [4096,256], "#ptr_dim_t" [128],
 * "#ingest_lut" uint16[768]. */
int ds2_model_create(const ds2_config* cfg, ds2_model** out);
void ds2_model_destroy(ds2_model* m);
int ds2_model_set_param(ds2_model* m, const char* name, const void* data, int64_t nbytes);
int ds2_model_finalize(ds2_model* m, void* stream);
/* A second execution context over the SAME weights (parameters, derived constants, bf16 weight planes are the parent's, which
 * must outlive the view); own workspace arena, GEMM scratch and arithmetic mode.  For running a stage on another stream
 * concurrently with the parent (the predictor encodes the next frames ahead of need this way).  A parent and its views may
 * be driven from different host threads (the shared weight-plane caches are filled under a mutex); ONE model or view is
 * not re-entrant - one host thread at a time, like the reference's predictor.  Destroy with ds2_model_destroy. */
int ds2_model_create_view(ds2_model* parent, ds2_model** out);

/* ---- A3: frame ingest.  load_video_frames, list-of-ndarray branch (sam2/utils/misc.py:280-284,
 * 328-342, 358-359): rgb_u8 [n,height,width,3] -> cv2.resize to S x S (8-bit INTER_LINEAR, OpenCV's fixed-point
 * algorithm; identity when already S x S) -> frames_f16 [n,3,S,S] = fp16((fp16(x/255) - mean) / std) with the
 * reference's fp16 roundings.  cv2 is absent from this image, so the resize step is restated from OpenCV's
 * published algorithm and is parity-unpinned (oracle/resize.py). */
int ds2_ingest_frames(ds2_model* m, const uint8_t* rgb_u8, int32_t n, int32_t height, int32_t width,
                      uint16_t* frames_f16, void* stream);

/* ---- A4+A5: SAM2Base.forward_image (sam2/modeling/sam2_base.py:450-461) = ImageEncoder.forward
 * (backbones/image_encoder.py:30-43): Hiera trunk (hieradet.py:283-299) + FpnNeck (:101-134, scalp=1)
 * + conv_s0/conv_s1 (mask_decoder.py:73-78).  frame_f16 [3,S,S] ->
 * fpn0 [65536,32], fpn1 [16384,64], fpn2 [4096,256]. */
int ds2_image_encoder(ds2_model* m, const uint16_t* frame_f16, float* fpn0, float* fpn1, float* fpn2, void* stream);
/* Same for n frames at once (frames_f16 [n,3,S,S] -> fpn0 [n,65536,32], fpn1 [n,16384,64], fpn2 [n,4096,256]):
 * the per-frame GEMMs of Hiera stages 3-4 (4096 / 1024 tokens) are too small to fill 256 CUs; the driver buffers
 * 30 frames before every pass (det_sam2_RT.py:429), so several can be encoded per launch. */
int ds2_image_encoder_batch(ds2_model* m, const uint16_t* frames_f16, int32_t n, float* fpn0, float* fpn1, float* fpn2,
                            void* stream);

/* ---- A11: memory-bank assembly, the tensor part of _prepare_memory_conditioned_features
 * (sam2_base.py:565-648).  feats[e]: bf16 [B,4096,64] of memory frame e, tpos_row[e] = index into
 * maskmem_tpos_enc (= num_maskmem - t_pos - 1); ptrs[i]: fp32 [B,256] object pointers with temporal
 * position ptr_pos[i] (already divided by max_obj_ptrs-1).  HOST arrays of device pointers.
 * memory / memory_pos: [B, Nk, 64], Nk = n_mem*4096 + 4*n_ptr. */
int ds2_bank_assemble(ds2_model* m, int32_t B, int32_t n_mem, const void* const* feats, const int32_t* tpos_row,
                      int32_t n_ptr, const float* const* ptrs, const float* ptr_pos, float* memory,
                      float* memory_pos, void* stream);

/* ---- A12: MemoryAttention.forward (sam2/modeling/memory_attention.py:119-176) with RoPEAttention
 * (sam/transformer.py:312-363).  curr [4096,256] is the level-2 feature of the frame, shared by the B
 * objects (the reference .expand()s it, sam2_video_predictor.py:1193-1206); curr_pos is the model
 * constant "#vision_pos".  out [B,4096,256].
 * Contract: the bank holds at least one memory frame - (Nk - num_obj_ptr_tokens) is a positive multiple of 4096 - in every
 * arithmetic mode (DS2_ERR_ARG otherwise).  A tracked frame of the reference always attends at least its conditioning
 * frame (sam2_base.py:565-590 adds the selected cond frames before anything else; a frame with no memory at all takes the
 * no_mem_embed branch :676-690 and never reaches memory_attention), so a pointer-only bank is not a state of the path. */
int ds2_memory_attention(ds2_model* m, int32_t B, const float* curr, const float* memory, const float* memory_pos,
                         int32_t Nk, int32_t num_obj_ptr_tokens, float* out, void* stream);

/* ---- A11 + A12 in one call, the tracking loop's form (sam2_base.py:500-690: _prepare_memory_conditioned_features builds the bank and
 * hands it to memory_attention): the entry tables of ds2_bank_assemble, curr as for ds2_memory_attention, out [B,4096,256].  Equal bit
 * for bit to ds2_bank_assemble followed by ds2_memory_attention; in mode bf16x3k with the assembly cross-attention the bank's entries
 * become the attention's operands directly (key-input planes, V^T tiles) and the fp32 memory / memory_pos [B,Nk,64] tensors are never
 * written (DS2_BANK_DIRECT=0: assembled in the workspace first, as in every other mode). */
int ds2_bank_memory_attention(ds2_model* m, int32_t B, const float* curr, int32_t n_mem, const void* const* feats,
                              const int32_t* tpos_row, int32_t n_ptr, const float* const* ptrs, const float* ptr_pos, float* out,
                              void* stream);

/* test hook: the cross-attention queries of memory-attention layer `layer` for fp32 rows x [rows,256] (rows % 64 == 0) in the assembly
 * attention's fragment order (rows/64 * 32 KiB): norm2 -> q_proj -> RoPE (row index modulo 4096) -> scale -> fp16, by the fused kernel
 * (fused != 0, gemm_qproj.hip) or by the three kernels it replaces. */
int ds2_op_query_fragments(ds2_model* m, int32_t layer, const float* x, int32_t rows, int32_t fused, void* qfrag, void* stream);

/* ---- A7+A8: SAM2Base._forward_sam_heads (sam2_base.py:254-397) = PromptEncoder.forward
 * (sam/prompt_encoder.py:134-171) + MaskDecoder.forward (sam/mask_decoder.py:105-161) + mask selection,
 * objectness gate, object pointer.  pix_feat: [B,4096,256], or [4096,256] shared when pix_bcast != 0
 * (init-cond frames; add_no_mem_embed != 0 adds no_mem_embed, sam2_base.py:651-657).  point_coords
 * [B,P,2] (1024-grid pixels) / point_labels [B,P] may be NULL (P=0: the reference's dummy point).
 * Outputs: low_res [B,256,256] logits of the selected mask, obj_ptr [B,256], obj_logits [B],
 * ious [B] (may be NULL). */
int ds2_sam_heads(ds2_model* m, int32_t B, const float* pix_feat, int32_t pix_bcast, int32_t add_no_mem_embed,
                  const float* fpn0, const float* fpn1, const float* point_coords, const int32_t* point_labels,
                  int32_t P, int32_t multimask_output, float* low_res, float* obj_ptr, float* obj_logits,
                  float* ious, void* stream);

/* ---- A7+A8 with a MASK prompt: same as ds2_sam_heads plus mask_inputs fp32 [B,256,256] (logits at the prompt
 * encoder's mask_input_size) whose PromptEncoder._embed_masks / mask_downscaling embedding
 * (prompt_encoder.py:60-68,97-100,163-168) replaces no_mask_embed as the dense prompt.  Used for a second prompt on
 * the same object and frame, where the reference feeds the previous prediction clamped to [-32,32] back in
 * (prev_sam_mask_logits, sam2_video_predictor.py:470-483; sam2_base.py:776-782).  mask_inputs == NULL = ds2_sam_heads. */
int ds2_sam_heads_mask(ds2_model* m, int32_t B, const float* pix_feat, int32_t pix_bcast, int32_t add_no_mem_embed,
                       const float* fpn0, const float* fpn1, const float* point_coords, const int32_t* point_labels,
                       int32_t P, const float* mask_inputs, int32_t multimask_output, float* low_res, float* obj_ptr,
                       float* obj_logits, float* ious, void* stream);

/* ---- A7 alone: PromptEncoder.forward (sam2/modeling/sam/prompt_encoder.py:134-171) as a module of its own.
 * point_coords [B,P,2] (1024-grid pixels; box corners are points labelled 2 / 3, :106-116) / point_labels [B,P]; pad != 0
 * appends the padding point (:81-85: what the reference does when `boxes is None`).  mask_inputs fp32 [B,256,256] or NULL
 * (-> no_mask_embed).  sparse [B, P + (P ? pad : 0), 256], dense [B,4096,256] token-major (either may be NULL). */
int ds2_prompt_encoder(ds2_model* m, int32_t B, const float* point_coords, const int32_t* point_labels, int32_t P, int32_t pad,
                       const float* mask_inputs, float* sparse, float* dense, void* stream);

/* ---- A8 alone: MaskDecoder.predict_masks (sam2/modeling/sam/mask_decoder.py:163-259) on the CALLER's prompt embeddings:
 * image_embeddings [B,4096,256], image_pe [4096,256], sparse [B,Ns,256], dense [B,4096,256] (all token-major), feat_s0
 * [65536,32] / feat_s1 [16384,64] = high_res_features (conv_s0 / conv_s1 applied) shared by the B objects ->
 * masks4 [B,4,256,256], iou4 [B,4], mask_tokens [B,4,256], obj_logits [B].  MaskDecoder.forward's slicing (:140-161:
 * multimask / dynamic stability selection) is done by the caller (det_sam2_amd.modules.HipMaskDecoder). */
int ds2_mask_decoder(ds2_model* m, int32_t B, const float* image_embeddings, const float* image_pe, const float* sparse,
                     int32_t Ns, const float* dense, const float* feat_s0, const float* feat_s1, float* masks4, float* iou4,
                     float* mask_tokens, float* obj_logits, void* stream);

/* ---- A13: SAM2Base._encode_new_memory (sam2_base.py:692-743) = 256->1024 bilinear upsample
 * (:355-360) + sigmoid|binarize, *20-10 + MemoryEncoder.forward (memory_encoder.py:158-181) +
 * no_obj_embed_spatial + bf16 storage.  fpn2 [4096,256] raw level-2 feature, low_res [B,256,256],
 * obj_logits [B] -> maskmem bf16 [B,4096,64]. */
int ds2_memory_encoder(ds2_model* m, int32_t B, const float* fpn2, const float* low_res, const float* obj_logits,
                       int32_t binarize, uint16_t* maskmem_bf16, void* stream);

/* ---- A14: sam2._C.get_connected_componnets (sam2/csrc/connected_components.cu:213-289; the reference's only native
 * op, called from sam2/utils/misc.py:48-61).  mask uint8 [N,1,H,W] (non-zero = foreground) -> labels int32
 * [N,1,H,W] (> 0 and equal exactly inside one 8-connected component, 0 on background; here the component's
 * smallest raster index + 1) and counts int32 [N,1,H,W] (component area per pixel, 0 on background).
 * work: int32 scratch of 2*N*H*W elements.  Any H, W (the reference requires even sizes). */
int ds2_connected_components(const uint8_t* mask, int32_t N, int32_t H, int32_t W, int32_t* labels, int32_t* counts,
                             int32_t* work, void* stream);

/* ---- A14: fill_holes_in_mask_scores (sam2/utils/misc.py:365-393; sam2_video_predictor.py:1343-1346): in place on
 * logits fp32 [N,1,H,W]: every 8-connected component of {logit <= 0} with area <= max_area gets logit 0.1.
 * work: int32 scratch of 3*N*H*W elements.  max_area <= 0 is an error (misc.py:371). */
int ds2_fill_holes(float* logits, int32_t N, int32_t H, int32_t W, int32_t max_area, int32_t* work, void* stream);

/* ---- F4, detector half: what ultralytics 8.2.82 does between the YOLOv8 head and `result.boxes`, which
 * VideoProcessor.detect_predict reads (det_sam2_RT.py:228-244): ops.non_max_suppression (best class per anchor, conf
 * threshold, xywh -> xyxy, per-class offset 7680, torchvision-style greedy NMS with IoU > iou_thres suppressing, max_det)
 * + scale_boxes / clip_boxes.  pred fp32 [nb, 4+nc, N] (device) -> dets fp32 [nb, max_det, 6] = x1, y1, x2, y2, conf, cls
 * in confidence order (ties: ascending anchor) and counts int32 [nb] (-1: more than 8192 anchors over the threshold).
 * scale5 (device, may be NULL) = gain, pad_x, pad_y, orig_w, orig_h of the letterbox to undo.  work: device scratch of
 * ds2_yolo_postprocess_work_bytes(nb, N) bytes.  Third-party algorithm, absent offline: parity unpinned (oracle/yolo_post.py). */
int64_t ds2_yolo_postprocess_work_bytes(int32_t nb, int32_t N);
int ds2_yolo_postprocess(const float* pred, int32_t nb, int32_t nc, int32_t N, float conf_thres, float iou_thres,
                         int32_t max_det, const float* scale5, float* dets, int32_t* counts, void* work, int64_t work_bytes,
                         void* stream);

/* ---- module-level forms (det_sam2_amd/modules.py: nn.Modules with the reference's signatures).  Same
 * kernels as the stage entry points above, without the tracking loop's sharing assumptions:
 *  ds2_image_encoder_f32: forward_image on fp32 frames [n,3,S,S] (the reference feeds `.float()`-ed frames,
 *    sam2_video_predictor.py:1186).
 *  ds2_memory_attention_ex: MemoryAttention.forward (memory_attention.py:119-176) with per-object tokens curr
 *    [B,4096,256] (curr_shared == 0) and an explicit curr_pos ([B,4096,256], or [4096,256] with pos_shared != 0; NULL =
 *    the model's "#vision_pos").
 *  ds2_memory_encoder_ex: MemoryEncoder.forward (memory_encoder.py:158-181): pix_feat [B,4096,256] (or shared
 *    [4096,256]), masks fp32 [B,1024,1024] (sigmoid applied unless skip_mask_sigmoid) -> vision_features fp32
 *    [B,4096,64] (token-major; no no_obj_embed_spatial, no bf16 rounding - those belong to _encode_new_memory). */
int ds2_image_encoder_f32(ds2_model* m, const float* frames_f32, int32_t n, float* fpn0, float* fpn1, float* fpn2, void* stream);
int ds2_memory_attention_ex(ds2_model* m, int32_t B, const float* curr, int32_t curr_shared, const float* curr_pos,
                            int32_t pos_shared, const float* memory, const float* memory_pos, int32_t Nk,
                            int32_t num_obj_ptr_tokens, float* out, void* stream);
int ds2_memory_encoder_ex(ds2_model* m, int32_t B, const float* pix_feat, int32_t pix_shared, const float* masks,
                          int32_t skip_mask_sigmoid, float* vision_features, void* stream);

/* ---- F3: mask prompts = add_new_mask (sam2/sam2_video_predictor.py:527-616) -> SAM2Base._use_mask_as_output
 * (sam2/modeling/sam2_base.py:399-448).  Three small ops the host composes with ds2_sam_heads_mask:
 *  ds2_resize_aa: F.interpolate(mode="bilinear", antialias=True, align_corners=False) of fp32 [B,Hin,Win] ->
 *    [B,Hout,Wout] (ATen's separable CPU algorithm, last dimension first); the source is mapped through
 *    in*in_scale + in_bias first; threshold < INFINITY binarises the result (out >= threshold ? 1 : 0).  Used for
 *    (i) a prompt mask given at another resolution -> 1024^2, >= 0.5 (sam2_video_predictor.py:552-561) and (ii) the
 *    low-res output logits: (mask*20-10) at 1024^2 -> 256^2 (sam2_base.py:407-415).  work: B*Hin*Wout floats.
 *  ds2_mask_prompt_prepare: mask_downsample Conv2d(1,1,4,stride 4) of the 0/1 mask [B,1024,1024] -> mask_ds [B,256,256]
 *    (the mask input of the SAM heads, sam2_base.py:425-429) and object_score_logits = any(mask > 0) ? +10 : -10
 *    (:436-440) -> obj_logits [B].  work: B int32.
 *  ds2_obj_ptr_gate: obj_ptr = lam*obj_ptr + (1-lam)*no_obj_ptr, lam = obj_logits > 0, in place on [B,256] (:441-444). */
int ds2_resize_aa(const float* in, int32_t B, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, float in_scale,
                  float in_bias, float threshold, float* work, float* out, void* stream);
int ds2_mask_prompt_prepare(ds2_model* m, int32_t B, const float* mask, float* mask_ds, float* obj_logits, int32_t* work,
                            void* stream);
int ds2_obj_ptr_gate(ds2_model* m, int32_t B, float* obj_ptr, const float* obj_logits, void* stream);

/* ---- A15: _get_orig_video_res_output (sam2_video_predictor.py:618-642) + `> 0` (det_sam2_RT.py:396-399):
 * low_res [B,256,256] -> logits fp32 [B,Hv,Wv] (may be NULL) and/or masks packed 8 px/byte, MSB first
 * (numpy.packbits rows, last byte zero-padded) [B,Hv,ceil(Wv/8)] (may be NULL). */
int ds2_mask_output(ds2_model* m, const float* low_res, int32_t B, int32_t Hv, int32_t Wv, float* logits,
                    uint8_t* packed, void* stream);

/* ---- arithmetic mode of the matrix-core kernels.  The mode is a property of a MODEL (ds2_model_set_precision; two models
 * in one process may differ); ds2_set_precision sets the process default that new models start with and that the
 * model-less primitive ops (ds2_op_*) use:
 *  0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
 *  1 = split-precision bf16x3: x = x0 + x1 in bf16, a0b0 + a0b1 + a1b0 with fp32 accumulation, ~2^-16 relative error
 *      per product;
 *  2 = bf16x3k (default): as 1, except on the tracking chain:
 *      - inside the memory attention (cross and self attention, sam/transformer.py:312-363) every operand of a matrix product
 *        is ONE IEEE fp16 plane (11-bit significand; queries after RoPE and scaling, keys after RoPE, the softmax weights
 *        - computed and summed in fp32 - and the values): q.k = q0 k0 and P.V = p0 v0 with fp32 accumulation, 1 MFMA term
 *        each (v_mfma_f32_32x32x16_f16).  The value operand keeps a second plane only for the object-pointer tokens (the
 *        frame tokens are bf16 storage in the bank, exactly representable in one fp16 plane);
 *      - the fused MLPs of the memory attention (linear1 / linear2, memory_attention.py:90-95) and of the memory encoder's
 *        CXBlocks (memory_encoder.py:104-117) use two fp16 terms per product: activations one fp16 plane, weights two
 *        (a0 w0 + a0 w1).
 *      Everything else (image encoder, SAM heads, projections, convolutions) is the three-term bf16 product of mode 1.
 *      Compared with mode 1 this removes two thirds of the score MFMAs, a third of the P.V MFMAs and the key lo plane from
 *      HBM and LDS.  Passes the precision gate recorded in DESIGN.md (every reference golden <= 1e-3 in 1 - IoU; measured
 *      <= 2.0e-4 at video resolution, tests/test_hip_measured_shape.py).
 * Softmax, LayerNorm, residuals and all storage stay fp32 in every mode. */
int ds2_set_precision(int32_t mode);
int ds2_get_precision(void);
int ds2_model_set_precision(ds2_model* m, int32_t mode);
int ds2_model_get_precision(const ds2_model* m);   /* -1 for a NULL model */

/* ---- measurement: HIP-event brackets (on the caller's stream) around named launch sites.  Tags:
 * "kernel.cross_attention", "kernel.self_attention", "stage.image_encoder", "stage.memory_attention",
 * "stage.sam_heads", "stage.memory_encoder".  ds2_profile_read waits for the recorded events, returns the
 * summed duration and launch count since the last read, and clears them. */
int ds2_profile_enable(int32_t on);
int ds2_profile_read(const char* tag, double* total_ms, int64_t* launches);
/* Newline-separated list of the tags that hold records (HOST buffer).  ds2_profile_enable(2) additionally brackets
 * every GEMM under the tag "gemm M N K" (the per-shape table behind bench.py's roofline_gemm). */
int ds2_profile_tags(char* buf, int64_t cap);

/* ---- primitive ops (exported for unit tests and for integrators who want a single op) */
int ds2_op_gemm(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* W, int32_t ldw,
                const float* bias, float* C, int32_t ldc, int32_t act, const float* gamma, const float* R,
                int32_t ldr, int32_t r_mod, void* stream);
/* The same GEMM with the result returned as bf16 operand planes (hi, lo: [M, round-up-32(N)] uint16 each, pad columns zero) instead of
 * fp32 - the form the first Linear layer of the Hiera MLP hands to the second (hieradet.py:160-166 via sam2_utils.py MLP); split
 * arithmetic modes only.  In mode 2 (bf16x3k), at a shape the MX product takes (M % 128 == 0, N % 192 == 0, K % 64 == 0, K >= 576, GELU,
 * 16-byte aligned bias / planes), the product is the two-MFMA-equivalent one and the planes are MX ACTIVATION planes: hi = IEEE fp16
 * (saturating), lo = one 16-bit word per element, byte 0 = e4m3(v * 4), byte 1 = e4m3((v - fp16(v)) * 2^14) (OCP e4m3, saturating). */
int ds2_op_gemm_planes(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* W, int32_t ldw,
                       const float* bias, int32_t act, uint16_t* out_hi, uint16_t* out_lo, void* stream);
/* x [rows, cols] fp32 (row stride ldx) -> its operand planes p1, p2 [rows, round-up-32(cols)]: fmt 0 = bf16 hi / lo; 1 = MX activation
 * planes (above); 2 = MX weight planes (p1 fp16; p2: byte 0 = e4m3((w - fp16(w)) * 2^18), byte 1 = e4m3(w * 64)); 3 = fp16 hi / lo.
 * The formats the producers inside the library emit for their consumer GEMMs; exported as a test hook. */
int ds2_op_split_planes(const float* x, int32_t ldx, int32_t rows, int32_t cols, int32_t fmt, uint16_t* p1, uint16_t* p2, void* stream);
/* fused two-layer MLP of width 256: out = (act(X W1^T + b1) W2^T + b2) * gamma + R with X [rows,256], W1 [H,256], W2 [256,H],
 * R / out [rows,256]; b1, b2, gamma, R may be NULL.  The kernel behind MemoryAttentionLayer's FFN (memory_attention.py:93-98)
 * and CXBlock's pwconv1 / pwconv2 (memory_encoder.py:104-117) in the bf16x3 modes. */
int ds2_op_mlp(int32_t rows, int32_t H, const float* X, const float* W1, const float* b1, const float* W2, const float* b2,
               const float* gamma, const float* R, float* out, int32_t act, void* stream);
/* Linear layer with 1 <= M <= 128 rows in exact fp32 (gemm_skinny.hip): C[M,N] = act(A W[N,K]^T + bias) * gamma + R.  The kernel
 * behind the token side of the two-way transformer (sam/transformer.py:139-236: 8 tokens per object) and the other few-row
 * Linear layers of ds2_sam_heads in every arithmetic mode; K, lda multiples of 4, A 16-byte aligned. */
int ds2_op_linear_small(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* W, int32_t ldw,
                        const float* bias, float* C, int32_t ldc, int32_t act, const float* gamma, const float* R,
                        int32_t ldr, int32_t r_mod, void* stream);
int ds2_op_layernorm(const float* x, const float* w, const float* b, float* y, int32_t rows, int32_t C, float eps,
                     int32_t act, void* stream);
int ds2_op_attention(const float* q, const float* k, const float* v, float* o, int32_t ldq, int32_t ldk, int32_t ldv,
                     int32_t ldo, int32_t batch, int32_t heads, int32_t D, int32_t DV, int32_t Lq, int32_t Lk,
                     float scale, int32_t win_q, int32_t win_k, int32_t Hq, int32_t Wq, int32_t Hk, int32_t Wk,
                     int32_t nwx, const float* k_pad, const float* v_pad, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DETSAM2_HIP_H */
