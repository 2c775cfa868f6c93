// Launchers of the HBM-bound / glue kernels (kernels.hip).  All asynchronous on `st`.
// Activations are fp32, token-major ("NHWC"): [rows, C] with C contiguous.
#pragma once
#include "common.h"

// Mode bf16x3k: the single-plane operands of the memory attention (queries, keys, softmax weights, values) as IEEE fp16 (1,
// default since round 4) or bf16 (0: the round-2/3 definition, kept for A/B builds: tools/ab.py build kbf16 -DDS2_ATTN_K_F16=0)
#ifndef DS2_ATTN_K_F16
#define DS2_ATTN_K_F16 1
#endif

int launch_add_bcast(const float* a, int lda, const float* b, int ldb, int b_mod, float alpha, float* out, int ldo,
                     int rows, int C, hipStream_t st);  // out[r,c] = a[r,c] + alpha*b[r % b_mod, c]  (b_mod<=0: r)
int launch_add_rowvec(const float* a, int lda, const float* vec, float* out, int ldo, int rows, int C, hipStream_t st);
int launch_maxpool2x2(const float* in, int ld_in, float* out, int ld_out, int H, int W, int C, hipStream_t st);
int launch_up2_add(const float* lat, const float* coarse, float* out, int H, int W, int C, hipStream_t st);
int launch_rope(float* x, int ldx, const float* cis, int batch, int L, int n_rope, int grid_tokens, hipStream_t st);
int launch_im2col_patch(const uint16_t* frame_f16, float* out, int S, hipStream_t st);  // [3,S,S] fp16 -> [(S/4)^2,148]
int launch_im2col_patch_f32(const float* frame_f32, float* out, int S, hipStream_t st);
int launch_ingest_u8(const uint8_t* rgb, const uint16_t* lut, uint16_t* out, int n, int S, hipStream_t st);
int launch_ingest_resize_u8(const uint8_t* rgb, const int* tab, const uint16_t* lut, uint16_t* out, int n, int H, int W, int S,
                            hipStream_t st);
int launch_permute4(const float* in, float* out, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3,
                    hipStream_t st);
int launch_pad_cols(const float* in, int rows, int cols, float* out, int cols_out, hipStream_t st);

// memory encoder
int launch_mask_upsample_transform(const float* low, float* high, int B, int hin, int hout, int mode, float scale,
                                   float bias, hipStream_t st);  // mode 0: sigmoid, 1: binarize(>0), 2: identity
int launch_conv3x3s2_small(const float* in, const float* w, const float* bias, const float* lnw, const float* lnb,
                           float* out, int B, int Hin, int Cin, int Cout, hipStream_t st);  // + LN2d(1e-6) + GELU
int launch_im2col3x3s2(const float* in, float* out, int B, int Hin, int Cin, hipStream_t st);
int launch_dwconv7(const float* in, const float* w49c, const float* bias, float* out, int B, int H, int C, hipStream_t st);
int launch_memfeat_finish(const float* feat, const float* obj_logits, const float* no_obj_embed, uint16_t* out_bf16,
                          int B, int tokens, int C, hipStream_t st);

// SAM heads
// keys[b,tok,:] = src[(b,)tok,:] + mask_downscaling(mask[b]) ; prm = {w0,b0,ln1w,ln1b,w3,b3,ln4w,ln4b,w6,b6}
int launch_mask_downscale_add(const float* mask, const float* const* prm, const float* src, int src_bcast, float* keys, int B,
                              hipStream_t st);
struct Mlp3Job {
  const float *A, *w0, *b0, *w1, *b1, *w2, *b2;
  float* out;
  int lda, n_out, ldc, last_act;
};
struct Mlp3Batch { Mlp3Job job[8]; };
int launch_mlp3_256_batch(const Mlp3Batch& jb, int n_jobs, int rows, hipStream_t st);   // several 3-layer MLPs, one launch
int launch_mlp3_256(const float* A, int lda, const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                    const float* b2, int n_out, float* out, int ldc, int last_act, int rows, hipStream_t st);
int launch_prompt_tokens(const float* out_tokens6, const float* gauss, const float* point_emb4, const float* not_a_point,
                         const float* coords, const int* labels, int B, int P, float image_size, float* tokens,
                         hipStream_t st, int pad = 1, const float* sparse_in = nullptr);  // tokens [B, 6+P+pad, 256]
int launch_upscale1(const float* g1, const float* feat_s1, const float* lnw, const float* lnb, float* u1, int B,
                    hipStream_t st, void* hi = nullptr, void* lo = nullptr);   // hi / lo: operand planes [B*16384, 64] instead of u1
int launch_upscale2_masks(const float* g2, const float* feat_s0, const float* hyper, float* masks, int B, hipStream_t st);
int launch_gather_rows(const float* in, int ld_in, int row_stride, int row_off, float* out, int ld_out, int B, int C,
                       hipStream_t st);  // out[b,:] = in[(b*row_stride+row_off), :]
int launch_select_masks(const float* masks4, const float* iou4, const float* obj_logits, const float* tokens_out,
                        int tok_ld, int multimask, float delta, float thresh, float* low_res, float* sel_token,
                        float* iou_out, int B, hipStream_t st);
int launch_ptr_gate(float* ptr, const float* obj_logits, const float* no_obj_ptr, int B, int C, hipStream_t st);

// mask prompts (add_new_mask): antialiased bilinear resize (work: [B,Hin,Wout] floats) and the mask_downsample conv
int launch_resize_aa(const float* in, float* work, float* out, int B, int Hin, int Win, int Hout, int Wout, float in_scale,
                     float in_bias, float thresh /* INFINITY = none */, hipStream_t st);
int launch_mask_downsample4(const float* mask, const float* w16, const float* bias, float* out, int* any /*[B]*/,
                            float* obj_logits /*[B]*/, int B, int S, hipStream_t st);

// memory bank.  The entry tables travel by value in the kernel arguments, at most DS2_MAX_*_ENTRIES per launch; larger
// banks (the reference has no limit: 20 selected cond frames + EVERY preload cond frame + 6, sam2_utils.py:56-60) are
// assembled by several launches, each writing its own slice of the [B, Nk, 64] outputs (e0 / p0 offsets).
#define DS2_MAX_MEM_ENTRIES 40
#define DS2_MAX_PTR_ENTRIES 40
struct BankArgs {
  int B, n_mem, n_ptr, tokens;             // entries in THIS launch; tokens = 4096
  int Nk, n_mem_total, e0, p0;             // whole bank: Nk = n_mem_total*tokens + 4*n_ptr_total; first entry / pointer of this launch
  const uint16_t* feats[DS2_MAX_MEM_ENTRIES];  // bf16 [B, tokens, 64]
  int tpos_row[DS2_MAX_MEM_ENTRIES];           // row of maskmem_tpos_enc to add
  const float* ptrs[DS2_MAX_PTR_ENTRIES];      // fp32 [B, 256]
  float ptr_pos[DS2_MAX_PTR_ENTRIES];          // signed temporal distance (already divided by t_diff_max)
  const float* maskmem_pos;                    // [tokens, 64]
  const float* tpos_enc;                       // [7, 64]
  const float* tpos_w; const float* tpos_b;    // obj_ptr_tpos_proj [64,256],[64]
  float* mem; float* mem_pos;                  // [B, Nk, 64] each, Nk = n_mem*tokens + 4*n_ptr
};
int launch_bank_assemble(const BankArgs& a, hipStream_t st);
int launch_bank_ptr(const BankArgs& a, const float* dim_t /*[128]*/, hipStream_t st);
// the bank straight to the cross-attention's operands (kin planes [B*Nk][64] bf16 hi / lo; V^T tiles of the assembly attention)
int launch_bank_kin(const BankArgs& a, void* hi, void* lo, hipStream_t st);
int launch_bank_vt32(const BankArgs& a, void* vt32, hipStream_t st);                       // (attention_x4a.hip: frame tokens)
int launch_bank_ptr_planes(const BankArgs& a, const float* dim_t, void* hi, void* lo, void* vt32, int ntile,
                           const unsigned char* vt_slot /*device [32]: key & 31 -> slot*/, hipStream_t st);
const unsigned char* attention_x4a_vt_slot_table();                                        // device table of vt_pos32
// norm2 + q_proj + RoPE + scale + fp16 pack of the memory cross-attention's queries in one kernel, straight into the assembly attention's
// Q fragments (gemm_qproj.hip); nrep > 1: rows = Lq shared queries written for nrep objects
int launch_x4a_qprep(const float* q, int ldq, int batch, int Lq, bool q_shared, float scale, const float* cis, int rope_grid, void* qfrag,
                     hipStream_t st);
// in_proj of the memory self-attention + key rotation / fp16 plane + V^T tiles in one kernel (gemm_qkvs.hip)
bool qkv_self_supported(int rows, int ldx, int ldw, int tokens_per_image);
int launch_qkv_self(const void* x_hi, const void* x_lo, int ldx, int rows, const void* w_hi, const void* w_lo, int ldw, const float* bias,
                    const float* cis, int rope_grid, float* q, int ldq, void* k_f16, void* vt, hipStream_t st);
bool qproj_x4a_supported(int rows, int ldx, int ldw);
int launch_qproj_x4a(const float* x, int ldx, int rows, const float* ln_w, const float* ln_b, float ln_eps, const void* w_hi, const void* w_lo,
                     int ldw, const float* bias, const float* cis, int rope_grid, float scale, void* qfrag, int nrep, hipStream_t st);

// outputs
int launch_mask_output(const float* low, int B, int hin, int Hv, int Wv, float* logits /*nullable*/,
                       uint8_t* packed /*nullable*/, hipStream_t st);

// producer of pre-split key planes for the memory attention (attention_w8.hip)
int launch_rope_split(const float* x, int ldx, const float* cis, int batch, int L, int n_rope, int grid_tokens,
                      void* hi, void* lo, hipStream_t st, bool hi_f16 = false);   // rows [batch*L] x 256 cols -> bf16 planes [batch*L][256]
                                                                                    // (hi_f16: the hi plane as fp16, mode bf16x3k)

// pre-split bf16x3 GEMM (gemm_split.hip)
struct GemmSplitArgs {
  int M, N, Kp;                                   // Kp = K rounded up to 32 (pad columns are zero)
  const unsigned short *A_hi, *A_lo; int lda;     // bf16 planes [M, lda]   (lda in elements, multiple of 8)
  const unsigned short *W_hi, *W_lo; int ldw;     // bf16 planes [N, ldw]
  const float* bias;
  float* C; int ldc;                              // fp32 result (may be null if only planes are wanted)
  int act; const float* gamma; const float* R; int ldr; int r_mod;
  unsigned short *C_hi, *C_lo; int ldcp;          // optional split planes of the result [M, ldcp] (pad cols zeroed)
  // optional axial RoPE applied to the result before it is split into planes (cross-attention keys): row m is
  // token t = m % rope_L of its batch item; tokens t < rope_n are rotated with cis[(t % rope_grid)][col/2]
  const float* rope_cis; int rope_L, rope_n, rope_grid;
  // axial table (compute_axial_cis): pairs [0, 64) depend on x = pos % rope_w only, pairs [64, 128) on y = pos / rope_w only.
  // rope_w > 0 lets a kernel read row x resp. row y * rope_w instead of row pos (identical values): the rows touched shrink
  // from rope_grid (4 MiB of table, competing with the streamed operands for L2) to 2 * rope_w (64 KiB).
  int rope_w;
  // block -> tile order inside an XCD's share of the grid: 0/1 = row-major over n; > 1 = groups of group_m tile rows
  // walked column-major (the ~32 blocks an XCD runs at once then share A rows AND W rows through its L2).  Set by
  // launch_gemm_split.
  int group_m;
  // k_gemm_split_d256: L2 prefetch distance of the A operand in K tiles (0 = off).  Set by launch_gemm_split.
  int prefetch;
  // C_hi is written as IEEE fp16 (11 significant bits) instead of bf16: the single-plane keys of the memory attention in
  // mode bf16x3k (attention_w8.hip).  Honoured by the K = 64 streaming kernel only (launch_gemm_split checks).
  int c_hi_f16;
  // the operand planes are "MX" planes (common.h: fp16 + fp8 byte pairs) and the product is the two-MFMA-equivalent one: taken by
  // the assembly kernel's 128 x 192 configuration only (gemm_x4g.hip "23m"; launch_gemm_split checks).  c_mx: the result planes
  // (C_hi / C_lo) are written as MX ACTIVATION planes for an MX consumer.
  int mx, c_mx;
};
int launch_gemm_split(const GemmSplitArgs& g, hipStream_t st);
int launch_gemm_split_r3(const GemmSplitArgs& g, hipStream_t st);           // 256x128 blocks, 8 waves, 3-stage LDS-DMA ring
int launch_gemm_split_d256(const GemmSplitArgs& g, hipStream_t st);         // 256x256, two-stage LDS-DMA, one barrier per K tile
int launch_gemm_split_pp256(const GemmSplitArgs& g, hipStream_t st);        // persistent p256 with loader / storer waves (no residual)
bool gemm_split_pp256_supported(const GemmSplitArgs& g);
// gemm_x4g.hip: assembly persistent kernel, epilogue of tile i under the main loop of tile i+1 (cfg 42: 256x128, 23: 128x192, 0: auto)
int gemm_split_x4g_config(const GemmSplitArgs& g, int cfg);                 // configuration that can take the GEMM, 0 if none
bool gemm_split_x4g_supported(const GemmSplitArgs& g);
int gemm_x4g_ncu();                                                         // CUs of the current device (the persistent grid)
int launch_gemm_split_x4g(const GemmSplitArgs& g, int cfg, hipStream_t st, const char** kname);
bool gemm_split_k64_supported(const GemmSplitArgs& g);                      // K = 64, N in {128, 256}, many rows
int launch_gemm_split_k64(const GemmSplitArgs& g, hipStream_t st);          // weight-stationary persistent streaming kernel
int launch_split_rows(const float* x, int ldx, int rows, int cols, void* hi, void* lo, int ldp, hipStream_t st,
                      bool f16 = false, int mx = 0);   // f16: IEEE fp16 planes instead of bf16; mx = DS2_PLANES_MX_A / _W: common.h
int launch_planes_bf16_to_mx(const void* hi, const void* lo, void* p1, void* p2, size_t elems, hipStream_t st);
// gemm_skinny.hip: fp32 linear layer for M <= 128 rows over the transposed weight Wt[K,N]
struct SkinnyArgs {
  int M, N, K;
  const float* A; int lda;
  const float* Wt;
  const float* bias; const float* gamma;
  const float* R; int ldr; int r_mod;
  float* C; int ldc; int act;
};
int launch_skinny_linear(const SkinnyArgs& g, hipStream_t st);
int launch_mask_up_conv1(const float* low, int hin, int Hin, int mode, float scale, float mbias, const float* w, const float* bias,
                         const float* lnw, const float* lnb, float* out, int B, hipStream_t st);
int launch_im2col3x3s2_split(const float* in, void* hi, void* lo, int ldp, int B, int Hin, int Cin, hipStream_t st);
int launch_bcast_rows(const float* x, float* out, int n, int B, hipStream_t st);
int launch_transpose_w(const float* W, int ldw, int N, int K, float* Wt, hipStream_t st);

// fused two-layer MLP, model width 256 (gemm_mlp256.hip): out = (act(X W1^T + b1) W2^T + b2) * gamma + R; the hidden
// activations stay in registers.  W2 planes must be built from launch_mlp256_permute_w2's output.
struct MlpArgs {
  int rows, D, H;                                  // D = 256; H a multiple of 128
  const unsigned short *X_hi, *X_lo; int ldx;      // bf16 planes [rows, ldx]
  const unsigned short *W1_hi, *W1_lo; int ldw1;   // [H, ldw1]
  const float* b1;                                 // [H] (nullable)
  const unsigned short *W2_hi, *W2_lo; int ldw2;   // [D, ldw2], hidden index permuted inside groups of 16
  const float* b2; const float* gamma;             // [D] (nullable)
  const float* R; int ldr;                         // residual fp32 [rows, ldr] (nullable)
  float* out; int ldo;                             // fp32 [rows, ldo]
  int act;                                         // DS2_ACT_NONE | RELU | GELU
  unsigned short *out_hi, *out_lo; int ldop;       // optional: the result ALSO as bf16 operand planes [rows, ldop] (next GEMM's A)
  // two-term fp16 products (mode bf16x3k, memory attention / memory encoder): W1 / W2 planes are FP16 (launch_split_rows
  // f16), X (read from its bf16 planes) and the hidden activations are rounded to ONE fp16 plane: x_h w_h + x_h w_l
  int f16x2;
  // fused LayerNorm of the RESULT rows (round 4: the next layer's norm1 / the final norm of the memory attention): when ln_w is
  // set, out_hi / out_lo receive the planes of LN(result) instead of the result's, and ln_out (nullable) its fp32 values;
  // `out` still receives the un-normalised result (the residual stream)
  const float *ln_w, *ln_b; float ln_eps; float* ln_out; int ldln;
  // few rows (fewer 128-row blocks than half of the CUs: few objects): the hidden dimension is split over hsplit workgroups per
  // row block; each writes its raw partial result to part [hsplit][rows][256] and k_mlp256_merge adds them up and applies the
  // epilogue (bias, gamma, residual, LayerNorm, planes).  Set by launch_mlp256 when `part` (scratch) is provided.
  float* part; size_t part_bytes; int hsplit;
  // fused LayerNorm of the INPUT rows (round 5: norm3 of the memory attention): when lni_w is set, X is read as fp32 rows X_f32 [rows, ldxf]
  // (X_hi / X_lo unused) and normalised in the kernel's prologue - the statistics in k_layernorm_vec's association order, the result
  // split into bf16 planes in registers and rounded to the fp16 plane exactly as the planes of a separate LayerNorm pass would be
  // (bit-identical); two-fp16-term ReLU / GELU forms
  const float* X_f32; int ldxf; const float *lni_w, *lni_b; float lni_eps;
};
bool mlp256_supported(const MlpArgs& a);
int launch_mlp256_permute_w2(const float* w2, int ldw, int n_rows, int H, float* out, hipStream_t st);
int launch_mlp256(const MlpArgs& a, hipStream_t st);
int mlp256_hsplit(int rows, int H, int ncu);       // parts of the hidden dimension a launch would be split into (1: none)
size_t mlp256_part_bytes(int rows, int H);         // scratch (MlpArgs::part) that split needs, 0 if none

// 8-wave variant on v_mfma_f32_16x16x32_bf16 (attention_w8.hip); V^T tiles use a different key permutation
// n_exact_keys/flag (optional): the leading n_exact_keys keys are expected to be bf16-exact; *flag (device int) is
// zeroed and then raised by the kernel if any of them has a non-zero lo part (see launch_attention_w8)
int launch_vt_split16(const float* v, int ldv, int batch, int L, void* vt, int dv, hipStream_t st, int n_exact_keys = 0,
                      int* flag = nullptr, bool f16 = false);   // dv = 64 | 128 | 256; f16: fp16 planes (mode bf16x3k)
int launch_attention_w8(const float* q, int ldq, const void* k_hi, const void* k_lo, const void* vt, float* o, int ldo,
                        int batch, int Lq, int Lk, float scale, int dv, hipStream_t st, void* o_hi = nullptr,
                        void* o_lo = nullptr, int ldop = 0,    // o_hi/o_lo: emit bf16 planes [rows, ldop] instead of fp32
                        int n_exact_keys = 0, const int* vlo_flag = nullptr,    // dv = 64: keys < n_exact_keys have a zero
                        const float* q_rope_cis = nullptr, int q_rope_grid = 0,   // rotate the queries while loading them
                        const float* res = nullptr, int ldres = 0,                // fp32 output: o = res + attention
                        bool q_shared = false,                                    // every batch item reads q[0 .. Lq)
                        float* split_ws = nullptr, size_t split_ws_bytes = 0);    // scratch for the key split (few workgroups)
                                                                                // V lo plane unless *vlo_flag != 0

// assembly 4-wave / 64-queries-per-wave cross-attention of mode bf16x3k (attention_x4a.hip): fp16 key plane [batch*Lk][256]
// (GemmSplitArgs::c_hi_f16), V^T tiles from launch_vt_pack32, result as bf16 operand planes; ws: attention_x4a_ws_bytes()
bool attention_x4a_enabled();
bool attention_x4a_supported(int batch, int Lq, int Lk, int dv, bool planes_out);
size_t attention_x4a_ws_bytes(int batch, int Lq, int Lk);
int launch_vt_pack32(const float* v, int ldv, int batch, int L, void* vt, hipStream_t st);   // [batch][ceil(L/32)][64][32] fp16
int launch_attention_x4a(const float* q, int ldq, const void* k_f16, const void* vt32, int batch, int Lq, int Lk, float scale,
                         hipStream_t st, void* o_hi, void* o_lo, int ldop, const float* q_rope_cis, int q_rope_grid, bool q_shared,
                         void* ws, size_t ws_bytes, bool merge = true);
int attention_x4a_parts(void* ws, int batch, int Lq, int Lk, const float** part_o, const float** part_ml);   // -> number of key-split parts
// merge / normalisation of the assembly attention's part(s) + the folded value / output projection + residual in one kernel (gemm_vo.hip)
bool vo_merge_supported(int rows, int ldw);
int launch_vo_merge(const float* part_o, const float* part_ml, int nsplit, int rows, const void* w_hi, const void* w_lo, int ldw, const float* bias,
                    const float* R, int ldr, int r_mod, float* out, int ldo, hipStream_t st);

// producers that emit bf16x3 operand planes directly (no fp32 round trip, no k_split_rows pre-pass)
int launch_layernorm_add(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, const float* add, float* y2,
                         int rows, int C, float eps, hipStream_t st);   // y = LN(x), y2 = y + add
int launch_layernorm_add_split(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, const float* add,
                               int add_mod, void* hi, void* lo, int ldp, int rows, int C, float eps, hipStream_t st,
                               void* hi0 = nullptr, void* lo0 = nullptr);
int launch_sam_keys_init(const float* src, int src_mod, const float* vec, const float* pe, int pe_mod, float* keys, void* khi,
                         void* klo, void* phi, void* plo, int rows, hipStream_t st);
int launch_layernorm_split(const float* x, int ldx, const float* w, const float* b, void* hi, void* lo, int ldp, int rows,
                           int C, float eps, int act, hipStream_t st, int mx = 0);
int launch_add_bcast_split(const float* a, int lda, const float* b, int ldb, int b_mod, float alpha, void* hi, void* lo,
                           int ldp, int rows, int C, hipStream_t st);
