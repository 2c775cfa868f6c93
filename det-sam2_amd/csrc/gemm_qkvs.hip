// Self-attention inputs of a memory-attention layer in ONE kernel (mode bf16x3k):
//     qkv = x_hat Wqkv^T + b  (x_hat = norm1's operand planes, 256 -> 768, bf16x3)
//       q  -> fp32 rows [rows, 256]                                (the attention kernel rotates its queries while loading them)
//       k  -> RoPE -> ONE fp16 plane [rows, 256]                   (k_rope_split's output)
//       v  -> V^T tiles of the 8-wave attention, fp16 hi plane     (k_vt_split16<256, true>'s hi plane; its lo plane is not read in this mode)
// instead of k_gemm_split_pp256 (fp32 qkv [rows, 768] out) -> k_rope_split -> k_vt_split16: three launches, 500 MB of traffic per layer
// for 67 MB of input and 133 MB of output.  (MemoryAttentionLayer._forward_sa, memory_attention.py:59-66; RoPEAttention, transformer.py:312-363.)
//
// Same arithmetic, same order per element as the three kernels (the layer output was compared bit for bit with the
// three-kernel chain in round 5, profiles/HISTORY.md): the tile kernels' term order per 16-deep k-step (a_lo w_hi, a_hi w_lo, a_hi w_hi); + bias; k_rope_split's rotation in the form
// hipcc contracts it to; the saturating fp16 pack of both producers.
// Orientation per 64-column group of the 768 outputs (a wave owns 32 tokens; x_hat's fragments are the same registers either way):
//   q, v : activations as the A operand - accumulator lane = output column, registers = tokens: q rows leave as 128-byte segments, a V^T row
//          (one dv, 32 keys in the attention's key order vt_pos16) is two 16-byte pieces of one lane;
//   k    : weights as the A operand (rows permuted while staged, gemm_qproj.hip) - accumulator lane = token, registers = 8 consecutive dims:
//          the complex pairs of the rotation sit in one lane, a fragment is one 16-byte store.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

#ifndef DS2_KROPE_VARIANT     // contraction of the rotation's imaginary parts (first pair, second pair of a float4): 0..3, see the epilogue
#define DS2_KROPE_VARIANT 0
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SD = 256;            // model width (k)
constexpr int SBR = 128;           // token rows per workgroup
constexpr int SSLOT = 16384;       // one ring slot: hi plane at +0, lo plane at +8192; 64 rows of 128 bytes (64 k)
constexpr int SLO = 8192;
constexpr int SNS = 6;            // ring slots: the DMA runs 5 tiles ahead, the fragment reads one tile ahead of the MFMAs
constexpr int STILES = 48;         // 12 groups of 64 output columns x 4 k tiles of 64

__device__ __forceinline__ unsigned s_cvt_pk_f16(float a, float b) {   // v_cvt_pk_f16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{a, b}), f16x2));
}
__host__ __device__ inline int s_delta(int rho) { return 16 * (rho >> 4) + 8 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 1) + (rho & 3); }

struct QkvsArgs {
  const unsigned short *X_hi, *X_lo; int ldx; int rows;   // norm1's planes [rows, ldx] bf16
  const unsigned short *W_hi, *W_lo; int ldw;             // in_proj weight planes [768, ldw] bf16
  const float* bias;                                      // [768]
  const float* cis; int rope_grid;                        // RoPE table [rope_grid][128][2], tokens per image
  float* q; int ldq;                                      // fp32 [rows, ldq]
  unsigned short* k;                                      // fp16 plane [rows, 256]
  unsigned short* vt;                                     // V^T tiles [rows / 32][2 planes][256 dv][32 pos] fp16 (hi plane written)
};

__global__ __launch_bounds__(256, 1) void k_qkv_self(QkvsArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  float* bs = reinterpret_cast<float*>(lds + SNS * SSLOT);   // [768] bias
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  for (int i = tid; i < 768; i += 256) bs[i] = a.bias ? a.bias[i] : 0.f;

  const char* wh = reinterpret_cast<const char*>(a.W_hi);
  const char* wl = reinterpret_cast<const char*>(a.W_lo);
  // DMA pieces of a 64-row tile: wave w issues pieces 2 w, 2 w + 1 (8 rows x 128 B) of both planes.  q / v groups: LDS row R <- weight
  // row R; k groups: LDS row R <- weight row 32 (R >> 5) + delta(R & 31)
  unsigned off_n[2], off_p[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = (wave * 2 + j) * 8 + (lane >> 3);
    const unsigned ch = (unsigned)(((lane & 7) ^ ((R >> 1) & 7)) * 8);
    off_n[j] = ((unsigned)R * (unsigned)a.ldw + ch) * 2u;
    off_p[j] = ((unsigned)(32 * (R >> 5) + s_delta(R & 31)) * (unsigned)a.ldw + ch) * 2u;
  }
  const int sw = (l31 >> 1) & 7;
  // tile T = 4 g + kt: output columns 64 g .., k 64 kt ..; groups 4..7 are the keys; ring slot T % 6
#define S_DMA(T)                                                                                                          \
  {                                                                                                                       \
    unsigned char* base_ = lds + ((T) % SNS) * SSLOT;                                                                     \
    const unsigned o_ = (unsigned)((T) >> 2) * 64u * (unsigned)a.ldw * 2u + (unsigned)((T) & 3) * 128u;                  \
    const bool perm_ = ((T) >> 4) == 1;                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                       \
      const unsigned of_ = (perm_ ? off_p[j] : off_n[j]) + o_;                                                            \
      __builtin_amdgcn_global_load_lds(wh + of_, (lds_ptr)(base_ + (wave * 2 + j) * 1024), 16, 0, 0);                     \
      __builtin_amdgcn_global_load_lds(wl + of_, (lds_ptr)(base_ + SLO + (wave * 2 + j) * 1024), 16, 0, 0);               \
    }                                                                                                                     \
  }
  __syncthreads();   // bs

  const int nrb = (a.rows + SBR - 1) / SBR;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int tok0 = rb * SBR + wave * 32;            // this wave's 32 tokens = one key tile of the attention
    const int tok = tok0 + l31;
    const int tokc = tok < a.rows ? tok : a.rows - 1;
    // ---- x_hat fragments: (token = lane & 31, k = 16 s + 8 half .. + 7) - the A operand of q / v, the B operand of k
    bf16x8 xh[SD / 16], xl[SD / 16];
    {
      const u32x4* ph = reinterpret_cast<const u32x4*>(a.X_hi + (size_t)tokc * a.ldx + half * 8);
      const u32x4* pl = reinterpret_cast<const u32x4*>(a.X_lo + (size_t)tokc * a.ldx + half * 8);
#pragma unroll
      for (int s = 0; s < SD / 16; ++s) {
        xh[s] = __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(ph + s * 2));
        xl[s] = __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(pl + s * 2));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the ring's counted waits below assume only DMA in flight)
    S_DMA(0) S_DMA(1) S_DMA(2) S_DMA(3) S_DMA(4)
    const int t = tokc % a.rope_grid;
    // fragments of TWO tiles: tile T's MFMAs run on one set while tile T + 1's reads fill the other (one wave per SIMD: nobody else covers
    // the LDS latency, and hipcc waits for a whole set at once)
    bf16x8 fh[2][4][2], fl[2][4][2];
#define S_READ(T, BUF)                                                                                                    \
  {                                                                                                                       \
    const unsigned char* base_ = lds + ((T) % SNS) * SSLOT;                                                               \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                         \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                                     \
        const unsigned char* r_ = base_ + (b * 32 + l31) * 128 + (((s * 2 + half) ^ sw) << 4);                            \
        fh[BUF][s][b] = *reinterpret_cast<const bf16x8*>(r_);                                                             \
        fl[BUF][s][b] = *reinterpret_cast<const bf16x8*>(r_ + SLO);                                                       \
      }                                                                                                                   \
  }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile 0 (tiles 1..4 may stay in flight)
    __builtin_amdgcn_s_barrier();
    S_READ(0, 0)
    // the K loop of one 64-column group (4 tiles); KEYS: weights as the A operand.  The group loops below are NOT unrolled (unrolled over
    // all 12 groups hipcc hoists the epilogues' addresses and spills 160 registers)
    // step T: tile T + 1 landed (this wave's pieces: the count; everybody's: the barrier) -> the DMA of tile T + 5 into the slot of tile
    // T - 1 (its fragments were consumed before anybody reached this barrier) -> tile T + 1's fragment reads -> tile T's MFMAs
#define S_STEP(g, KT, KEYS)                                                                                               \
  {                                                                                                                       \
    const int T = (g) * 4 + (KT);                                                                                         \
    const int rem = STILES - 2 - T;   /* tiles behind T + 1 that exist */                                                  \
    if (rem >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                                       \
    else if (rem == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                   \
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                   \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                         \
    if (T + 5 < STILES) S_DMA(T + 5)                                                                                      \
    if (T + 1 < STILES) S_READ(T + 1, ((KT) + 1) & 1)                                                                     \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                         \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                                     \
        constexpr int ks = (KT) * 4;                                                                                      \
        constexpr int cb = (KT) & 1;                                                                                      \
        if (KEYS) {   /* weights as the A operand: a_lo w_hi, a_hi w_lo, a_hi w_hi */                                      \
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[cb][s][b], xl[ks + s], acc[b], 0, 0, 0);                    \
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[cb][s][b], xh[ks + s], acc[b], 0, 0, 0);                    \
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[cb][s][b], xh[ks + s], acc[b], 0, 0, 0);                    \
        } else {      /* activations as the A operand */                                                                  \
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[ks + s], fh[cb][s][b], acc[b], 0, 0, 0);                    \
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[ks + s], fl[cb][s][b], acc[b], 0, 0, 0);                    \
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[ks + s], fh[cb][s][b], acc[b], 0, 0, 0);                    \
        }                                                                                                                 \
      }                                                                                                                   \
  }
    // the K loop of one 64-column group (4 tiles, unrolled: x_hat's fragments and the fragment sets are indexed statically); KEYS: weights
    // as the A operand.  The group loops below are NOT unrolled (unrolled over all 12 groups hipcc hoists the epilogues' addresses and
    // spills 160 registers)
#define S_GROUP_MAIN(g, KEYS)                                                                                             \
  f32x16 acc[2];                                                                                                          \
  _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                           \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;                                                       \
  S_STEP(g, 0, KEYS) S_STEP(g, 1, KEYS) S_STEP(g, 2, KEYS) S_STEP(g, 3, KEYS)
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
      S_GROUP_MAIN(g, false)
      // ---- queries: lane = column 64 g + 32 b + l31, register e = token (e & 3) + 8 (e >> 2) + 4 half of the wave's 32
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = 64 * g + 32 * b + l31;
        const float bv = bs[col];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int tk = tok0 + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (tk < a.rows) a.q[(size_t)tk * a.ldq + col] = acc[b][e] + bv;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int g = 4; g < 8; ++g) {
      S_GROUP_MAIN(g, true)
      // ---- keys: registers 0..7 / 8..15 of block b are dims 16 ks + 8 half .. + 7, ks = 2 (2 (g - 4) + b) + f, of this lane's token
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int ks = 2 * (2 * (g - 4) + b) + f, d0 = 16 * ks + 8 * half;
          const float4 bb0 = *reinterpret_cast<const float4*>(bs + 256 + d0), bb1 = *reinterpret_cast<const float4*>(bs + 256 + d0 + 4);
          float4 v0 = make_float4(acc[b][8 * f + 0] + bb0.x, acc[b][8 * f + 1] + bb0.y, acc[b][8 * f + 2] + bb0.z, acc[b][8 * f + 3] + bb0.w);
          float4 v1 = make_float4(acc[b][8 * f + 4] + bb1.x, acc[b][8 * f + 5] + bb1.y, acc[b][8 * f + 6] + bb1.z, acc[b][8 * f + 7] + bb1.w);
          {   // k_rope_split: two complex pairs per float4, table row t, pair index d / 2
            const float4 c0 = *reinterpret_cast<const float4*>(a.cis + ((size_t)t * 128 + d0 / 2) * 2);
            const float4 c1 = *reinterpret_cast<const float4*>(a.cis + ((size_t)t * 128 + d0 / 2 + 2) * 2);
            auto re = [](float x, float y, float cx, float cy) { return __builtin_fmaf(x, cx, -(y * cy)); };
            auto imA = [](float x, float y, float cx, float cy) { return __builtin_fmaf(x, cy, y * cx); };
            auto imB = [](float x, float y, float cx, float cy) { return __builtin_fmaf(y, cx, x * cy); };
#if DS2_KROPE_VARIANT == 0
#define IM1 imA
#define IM2 imB
#elif DS2_KROPE_VARIANT == 1
#define IM1 imB
#define IM2 imA
#elif DS2_KROPE_VARIANT == 2
#define IM1 imA
#define IM2 imA
#else
#define IM1 imB
#define IM2 imB
#endif
            v0 = make_float4(re(v0.x, v0.y, c0.x, c0.y), IM1(v0.x, v0.y, c0.x, c0.y), re(v0.z, v0.w, c0.z, c0.w), IM2(v0.z, v0.w, c0.z, c0.w));
            v1 = make_float4(re(v1.x, v1.y, c1.x, c1.y), IM1(v1.x, v1.y, c1.x, c1.y), re(v1.z, v1.w, c1.z, c1.w), IM2(v1.z, v1.w, c1.z, c1.w));
          }
          auto pk = [](float x, float y) { return s_cvt_pk_f16(ds2_sat_f16(x), ds2_sat_f16(y)); };
          const uint4 o = make_uint4(pk(v0.x, v0.y), pk(v0.z, v0.w), pk(v1.x, v1.y), pk(v1.z, v1.w));
          if (tok < a.rows) *reinterpret_cast<uint4*>(a.k + (size_t)tok * 256 + d0) = o;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int g = 8; g < 12; ++g) {
      S_GROUP_MAIN(g, false)
      // ---- values: lane = dv 64 (g - 8) + 32 b + l31, register e = key (e & 3) + 8 (e >> 2) + 4 half of the tile; position of a key in the
      //      V^T row: vt_pos16 = 16 ((e >> 2) & 1) + 8 half + 4 (e >> 3) + (e & 3) - registers {0..3, 8..11} and {4..7, 12..15} are two pieces
      if (tok0 < a.rows) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int dv = 64 * (g - 8) + 32 * b + l31;
          const float bv = bs[512 + dv];
          auto pk = [&](int e) { return s_cvt_pk_f16(ds2_sat_f16(acc[b][e] + bv), ds2_sat_f16(acc[b][e + 1] + bv)); };
          unsigned short* row = a.vt + ((size_t)(tok0 >> 5) * 2 * (32 * 256)) + (size_t)dv * 32;
          *reinterpret_cast<uint4*>(row + 8 * half) = make_uint4(pk(0), pk(2), pk(8), pk(10));
          *reinterpret_cast<uint4*>(row + 16 + 8 * half) = make_uint4(pk(4), pk(6), pk(12), pk(14));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the ring restarts: nobody still reads the last tiles
  }
}

}  // namespace

bool qkv_self_supported(int rows, int ldx, int ldw, int tokens_per_image) {
  return rows > 0 && rows % 32 == 0 && ldx % 8 == 0 && ldw % 64 == 0 && tokens_per_image % 32 == 0;
}

int launch_qkv_self(const void* x_hi, const void* x_lo, int ldx, int rows, const void* w_hi, const void* w_lo, int ldw, const float* bias,
                    const float* cis, int rope_grid, float* q, int ldq, void* k_f16, void* vt, hipStream_t st) {
  DS2_REQUIRE(x_hi && x_lo && w_hi && w_lo && cis && q && k_f16 && vt && rope_grid > 0 && ldq % 4 == 0 && qkv_self_supported(rows, ldx, ldw, rope_grid),
              "qkv_self: bad argument");
  QkvsArgs a{};
  a.X_hi = reinterpret_cast<const unsigned short*>(x_hi); a.X_lo = reinterpret_cast<const unsigned short*>(x_lo); a.ldx = ldx; a.rows = rows;
  a.W_hi = reinterpret_cast<const unsigned short*>(w_hi); a.W_lo = reinterpret_cast<const unsigned short*>(w_lo); a.ldw = ldw;
  a.bias = bias; a.cis = cis; a.rope_grid = rope_grid; a.q = q; a.ldq = ldq;
  a.k = reinterpret_cast<unsigned short*>(k_f16); a.vt = reinterpret_cast<unsigned short*>(vt);
  static int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int nrb = (rows + SBR - 1) / SBR;
  const int grid = nrb < ncu ? nrb : ncu;
  const size_t smem = (size_t)SNS * SSLOT + 768 * 4;
  static bool attr_done = false;
  if (!attr_done) {
    DS2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_self), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(k_qkv_self, dim3(grid), dim3(256), smem, st, a);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
