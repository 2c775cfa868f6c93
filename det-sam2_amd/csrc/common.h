// Common declarations for the det-sam2 gfx950 kernels (internal; the public C-ABI is
// include/detsam2_hip.h).  Everything here targets CDNA4 / wave64 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#ifndef DS2_OK      /* (the same values as include/detsam2_hip.h) */
#define DS2_OK 0
#define DS2_ERR_ARG 1
#define DS2_ERR_HIP 2
#define DS2_ERR_STATE 3
#define DS2_ERR_UNSUPPORTED 4
#endif

void ds2_set_error(const char* fmt, ...);

#define DS2_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ds2_set_error("%s:%d HIP error %s in %s", __FILE__, __LINE__, hipGetErrorString(_e), #expr); \
      return DS2_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define DS2_CHECK_LAUNCH() DS2_CHECK_HIP(hipGetLastError())

#define DS2_REQUIRE(cond, ...)                     \
  do {                                             \
    if (!(cond)) {                                 \
      ds2_set_error(__VA_ARGS__);                  \
      return DS2_ERR_ARG;                          \
    }                                              \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Row of the 32x32 MFMA C/D fragment held in accumulator register r by a lane in half h
// (lane>>5): row = (r&3) + 8*(r>>2) + 4*h; the column is lane&31.
__device__ __forceinline__ int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// fp32 -> fp16 operand planes (mode bf16x3k) SATURATE: |x| > 65504 becomes +-65504, not inf - an out-of-range activation of a real
// checkpoint must degrade a product, not turn the frame into NaNs (bf16 planes have fp32's range and need nothing)
__device__ __forceinline__ float ds2_sat_f16(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }

// ---- "MX" operand planes of the two-MFMA-equivalent product (round 6; gemm_x4g.hip, tools/gen/gen_gemm_x4g.py "23m"):
//   a w ~= a16 w16  +  a8 wl8  +  al8 w8     a16 = fp16(a), w16 = fp16(w): v_mfma_f32_32x32x16_f16 (full rate)
//                                            both cross terms of 32 k's in ONE v_mfma_scale_f32_32x32x64_f8f6f4 (twice the rate):
//   plane 1 of an operand = fp16 [rows, ld]; plane 2 = one 16-bit word per element, same shape and addressing:
//     activation:  byte 0 = e4m3(a 2^-EA)                byte 1 = e4m3((a - a16) 2^LA)
//     weight:      byte 0 = e4m3((w - w16) 2^LW)         byte 1 = e4m3(w 2^-EW)
//   so that the byte pairs of an A row and a W row multiply to a8 wl8 + al8 w8, both with the factor 2^(EA - LW) = 2^(EW - LA)
//   (EA + LA = EW + LW), which the instruction's E8M0 scale operand applies.  The scales are STATIC powers of two - no block maxima
//   in any producer: a value outside e4m3's range saturates (|a| > 448 2^EA = 112, |a - a16| > 448 2^-LA, i.e. |a| > ~56; |w| > 7 resp.
//   3.5), which degrades THAT element's cross term to the plain fp16 product's error, and a value below the subnormal step loses
//   a term that is below 2^-21 of the row's largest product.  OCP e4m3 (v_cvt_pk_fp8_f32 on gfx950), round to nearest even.
#define DS2_MX_EA (-2)
#define DS2_MX_LA 14
#define DS2_MX_EW (-6)
#define DS2_MX_LW 18
static_assert(DS2_MX_EA + DS2_MX_LA == DS2_MX_EW + DS2_MX_LW, "both cross terms must carry the same power of two");
enum { DS2_PLANES_BF16 = 0, DS2_PLANES_MX_A = 1, DS2_PLANES_MX_W = 2 };   // format of an operand-plane pair
#ifdef __HIPCC__
// (x0, x1) -> low / high half of the plane-2 word pair; e = exponent of the value byte's scale, l = of the remainder byte's
__device__ __forceinline__ unsigned ds2_mx_word2(float x0, float r0, float x1, float r1, bool weight) {
  const float sv = weight ? __builtin_ldexpf(1.f, -DS2_MX_EW) : __builtin_ldexpf(1.f, -DS2_MX_EA);
  const float sr = weight ? __builtin_ldexpf(1.f, DS2_MX_LW) : __builtin_ldexpf(1.f, DS2_MX_LA);
  const float v0 = __builtin_amdgcn_fmed3f(x0 * sv, -448.f, 448.f), v1 = __builtin_amdgcn_fmed3f(x1 * sv, -448.f, 448.f);
  const float q0 = __builtin_amdgcn_fmed3f(r0 * sr, -448.f, 448.f), q1 = __builtin_amdgcn_fmed3f(r1 * sr, -448.f, 448.f);
  int w = 0;
  if (weight) {   // remainder byte first
    w = __builtin_amdgcn_cvt_pk_fp8_f32(q0, v0, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(q1, v1, w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v0, q0, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v1, q1, w, true);
  }
  return (unsigned)w;
}
// two adjacent values -> their plane-1 word (fp16 pair, saturating) and plane-2 word
__device__ __forceinline__ void ds2_mx_pair(float x0, float x1, bool weight, unsigned& p1, unsigned& p2) {
  typedef _Float16 mxh2 __attribute__((ext_vector_type(2)));
  typedef float mxf2 __attribute__((ext_vector_type(2)));
  const mxh2 h = __builtin_convertvector((mxf2{__builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f)}), mxh2);
  p1 = __builtin_bit_cast(unsigned, h);
  p2 = ds2_mx_word2(x0, x0 - (float)h[0], x1, x1 - (float)h[1], weight);
}
#endif

// ---- epilogue / activation codes shared by GEMM and LayerNorm
enum { DS2_ACT_NONE = 0, DS2_ACT_RELU = 1, DS2_ACT_GELU = 2, DS2_ACT_SIGMOID = 3 };

// GELU(x) = x * Phi(x), Phi(x) = erfc(-x / sqrt 2) / 2.  erfc(z), z >= 0, by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 absolute): branch-free, one v_exp_f32 and one v_rcp_f32 - the libm erff() costs ~3x the VALU work and
// the GEMM epilogues of the MLPs evaluate it once per output element.  The negative side uses the complementary form
// directly (no 1 - erf cancellation), so the tail keeps its relative accuracy.
__device__ __forceinline__ float ds2_gelu(float x) {
  // (contraction off + explicit fmaf: every kernel that inlines this must round identically - the tile choice of a GEMM
  // depends on its row count, and a stream sharded over ranks encodes other batch sizes than a sequential one)
#pragma clang fp contract(off)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));   // v_rcp_f32 (1 ulp); __frcp_rn is a 10-instruction IEEE division
  float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  p = __builtin_fmaf(t, p, 1.421413741f);
  p = __builtin_fmaf(t, p, -0.284496736f);
  p = __builtin_fmaf(t, p, 0.254829592f);
  const float poly = t * p;
  const float half_erfc = (0.5f * poly) * __expf(-(z * z));           // erfc(z) / 2
  const float r = x >= 0.f ? 1.f - half_erfc : half_erfc;
  return x * r;
}

__device__ __forceinline__ float ds2_act(float x, int act) {
  if (act == DS2_ACT_RELU) return x > 0.f ? x : 0.f;
  if (act == DS2_ACT_GELU) return ds2_gelu(x);
  if (act == DS2_ACT_SIGMOID) return 1.f / (1.f + expf(-x));
  return x;
}

// four outputs at once with the (wave-uniform) activation switch taken ONCE: the GEMM epilogues evaluate it per row of four
// columns, and a per-element switch cost them ~12 scalar branches per row (tools/k64_trace_bench.py: the epilogues are
// instruction-issue bound)
__device__ __forceinline__ void ds2_act4(float (&v)[4], int act) {
  if (act == DS2_ACT_NONE) return;
  if (act == DS2_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
  } else if (act == DS2_ACT_GELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ds2_gelu(v[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = 1.f / (1.f + expf(-v[j]));
  }
}

// ---- primitive launchers (implemented in the .hip files; all asynchronous on `st`)
struct GemmArgs {
  int M, N, K;            // C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) * gamma[N] + R[M,N]
  const float* A; int lda;
  const float* W; int ldw;
  const float* bias;      // may be null
  float* C; int ldc;
  int act;
  const float* gamma;     // may be null (per-column scale applied after act)
  const float* R; int ldr; // may be null; residual added last
  int r_mod;              // if >0 the residual row is (m % r_mod) (broadcast over a leading batch)
};
int launch_gemm(const GemmArgs& g, hipStream_t st);          // exact fp32 MFMA

// arithmetic mode of the matrix-core kernels (ds2_set_precision)
// BF16X3K = BF16X3 everywhere except the memory attention (cross + self): its SCORES are plain bf16 x bf16 products (keys and
// queries as one bf16 plane each, fp32 accumulation) and its softmax weights enter P.V as one bf16 plane - see DESIGN.md
// "precision margin"
enum { DS2_PREC_FP32 = 0, DS2_PREC_BF16X3 = 1, DS2_PREC_BF16X3K = 2 };
// The mode belongs to a ds2_model (ds2_model_set_precision); every model entry point installs it for the calling thread
// while it runs (ds2_precision_scope), so two models in one process - other GPUs, other streams, other modes - never
// see each other's setting.  Outside a model call (the primitive ops) the process default applies (ds2_set_precision).
extern int g_ds2_default_precision;
extern thread_local int t_ds2_precision;          // < 0: no model call in progress on this thread
static inline int ds2_precision() { return t_ds2_precision >= 0 ? t_ds2_precision : g_ds2_default_precision; }
static inline bool ds2_split_mode() { return ds2_precision() != DS2_PREC_FP32; }
struct ds2_precision_scope {
  int prev;
  explicit ds2_precision_scope(int mode) : prev(t_ds2_precision) { t_ds2_precision = mode; }
  ~ds2_precision_scope() { t_ds2_precision = prev; }
};

// per-kernel HIP-event brackets (ds2_profile_enable(2)): tag "kern <kernel> M N K", the kernel alone (no operand pre-pass)
bool ds2_prof_kernels();
void ds2_prof_record(const char* tag, hipEvent_t a, hipEvent_t b);   // takes ownership of the two recorded events

int launch_layernorm(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int rows, int C,
                     float eps, int act, hipStream_t st);

struct AttnArgs {
  const float *q, *k, *v; float* o;
  int ldq, ldk, ldv, ldo;       // row strides in floats
  int batch, heads, D, DV;      // q/k head dim D, v head dim DV; head h lives at column h*D (h*DV for v,o)
  int Lq, Lk;                   // tokens per batch item (for windowed mode: window area)
  float scale;                  // softmax scale (1/sqrt(D))
  // windowed mode (Hiera): batch item = window (wy*nwx + wx) of one image stored in natural (y,x) row order
  int win_q, win_k;             // 0 = plain [batch, L] rows
  int Hq, Wq, Hk, Wk, nwx;
  int wins;                     // windowed mode with several images: windows per image (0 = one image); image i's
                                // tokens are rows [i*H*W, (i+1)*H*W) and batch item = i*wins + window
  const float *k_pad, *v_pad;   // row used for padded key positions (the qkv bias), may be null
  // bf16x3 kernels only: if set, the result is written as two bf16 planes [rows, ldop] (GEMM operand format)
  // instead of fp32 `o`
  unsigned short *o_hi, *o_lo; int ldop;
  int o_mx;                     // the planes are "MX" activation planes (fp16 + fp8 byte pairs, see above) for an MX GEMM consumer
};
int launch_attention(const AttnArgs& a, hipStream_t st);         // dispatches on ds2_precision()
bool attention_fewq_supported(const AttnArgs& a);                // Lq <= 16 against >= 1024 keys, head dim 16/32
int launch_attention_fewq(const AttnArgs& a, hipStream_t st);    // split-key exact fp32 path (attention_fewq.hip)
bool attention_smallwin_supported(const AttnArgs& a);            // 16- / 64-key Hiera windows, bf16x3 (attention_smallwin.hip)
int launch_attention_smallwin(const AttnArgs& a, hipStream_t st);
bool attention_winlds_supported(const AttnArgs& a);               // 16 x 16 / 14 x 14 Hiera windows, head dim 72, whole window in LDS (attention_winlds.hip)
int launch_attention_winlds(const AttnArgs& a, hipStream_t st);
// Hiera global attention over operands pre-split per (image, head) (attention_hg.hip): the caller provides the plane buffers
bool attention_hg_supported(const AttnArgs& a);
size_t attention_hg_k_bytes(const AttnArgs& a);                  // K tile images (hi + lo planes)
size_t attention_hg_vt_bytes(const AttnArgs& a);                 // V^T tile images
int launch_attention_hg(const AttnArgs& a, void* k_img, void* vt_img, hipStream_t st);
int launch_attention_bf16x3(const AttnArgs& a, hipStream_t st);  // DS2_ERR_UNSUPPORTED if no kernel for (D,DV)
