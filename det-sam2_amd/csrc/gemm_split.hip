// bf16x3 GEMM over PRE-SPLIT operands:  C[M,N] = epi( (A0+A1)[M,K] * (W0+W1)[N,K]^T )
//
// Same arithmetic as gemm_bf16x3.hip (a0*w0 + a0*w1 + a1*w0, fp32 accumulate) but the fp32 -> (hi,lo) bf16
// split is NOT done in the tile loop: weights are split once per model (cached), activations once per tensor
// (k_split_rows, or directly by the producing kernel).  The tile loop is then 8 unconditional 16-byte global
// loads, 8 ds_write_b128, 16 ds_read_b128 and 24 v_mfma_f32_32x32x16_bf16 per wave - no conversion VALU.
// The epilogue can emit fp32 and/or the split planes of the result (for a consumer GEMM).
#include <stdlib.h>
#include <string.h>

#include <map>
#include <tuple>
#include <vector>

#include "common.h"
#include "kernels.h"

#ifndef DS2_GEMM_INTERLEAVE
#define DS2_GEMM_INTERLEAVE 1
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BM = 128, BN = 128, BK = 32, ROWB = 80, PLANE = BM * ROWB;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// fp32 [rows, cols] (row stride ldx) -> bf16 planes [rows, ldp] (cols..ldp zero padded); one thread / 4 cols
__global__ void k_split_rows(const float* x, int ldx, int rows, int cols, uint2* hi, uint2* lo, int ldp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = ldp / 4;
  if (i >= (size_t)rows * q) return;
  const int c4 = (int)(i % q);
  const size_t r = i / q;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < cols) v = *reinterpret_cast<const float4*>(x + r * ldx + c4 * 4);
  uint2 h, l;
  h.x = cvt_pk_bf16(v.x, v.y);
  h.y = cvt_pk_bf16(v.z, v.w);
  l.x = cvt_pk_bf16(v.x - bf_lo(h.x), v.y - bf_hi(h.x));
  l.y = cvt_pk_bf16(v.z - bf_lo(h.y), v.w - bf_hi(h.y));
  hi[i] = h;
  lo[i] = l;
}

// the same with IEEE fp16 planes (hi = fp16(x), lo = fp16(x - hi): 22 bits together; |x| < 65504).  Weights of the two-term
// fp16 products (gemm_mlp256.hip, mode bf16x3k)
__global__ void k_split_rows_f16(const float* x, int ldx, int rows, int cols, uint2* hi, uint2* lo, int ldp) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = ldp / 4;
  if (i >= (size_t)rows * q) return;
  const int c4 = (int)(i % q);
  const size_t r = i / q;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < cols) v = *reinterpret_cast<const float4*>(x + r * ldx + c4 * 4);
  const h2 h0 = __builtin_convertvector((f2{ds2_sat_f16(v.x), ds2_sat_f16(v.y)}), h2);
  const h2 h1 = __builtin_convertvector((f2{ds2_sat_f16(v.z), ds2_sat_f16(v.w)}), h2);
  const h2 l0 = __builtin_convertvector((f2{v.x - (float)h0[0], v.y - (float)h0[1]}), h2);
  const h2 l1 = __builtin_convertvector((f2{v.z - (float)h1[0], v.w - (float)h1[1]}), h2);
  hi[i] = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
  lo[i] = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}

// the same with the "MX" planes of the two-MFMA-equivalent product (common.h): WEIGHT selects the byte order of plane 2
template <bool WEIGHT>
__global__ void k_split_rows_mx(const float* x, int ldx, int rows, int cols, uint2* hi, uint2* lo, int ldp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = ldp / 4;
  if (i >= (size_t)rows * q) return;
  const int c4 = (int)(i % q);
  const size_t r = i / q;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < cols) v = *reinterpret_cast<const float4*>(x + r * ldx + c4 * 4);
  uint2 h, l;
  ds2_mx_pair(v.x, v.y, WEIGHT, h.x, l.x);
  ds2_mx_pair(v.z, v.w, WEIGHT, h.y, l.y);
  hi[i] = h;
  lo[i] = l;
}
// bf16 hi / lo planes (a = hi + lo to 2^-17) -> MX activation planes of the same shape (the operand of an MX GEMM whose producer
// emitted bf16 planes)
__global__ void k_planes_bf16_to_mx(const uint2* hi, const uint2* lo, uint2* p1, uint2* p2, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint2 h = hi[i], l = lo[i];
  uint2 a, b;
  ds2_mx_pair(bf_lo(h.x) + bf_lo(l.x), bf_hi(h.x) + bf_hi(l.x), false, a.x, b.x);
  ds2_mx_pair(bf_lo(h.y) + bf_lo(l.y), bf_hi(h.y) + bf_hi(l.y), false, a.y, b.y);
  p1[i] = a;
  p2[i] = b;
}

template <int DBG>
__global__ __launch_bounds__(256, 2) void k_gemm_split(GemmSplitArgs g, int mt, int nt) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][2][PLANE];   // [buf][A/B][hi/lo]

  const int nwg = mt * nt;
  const int orig = blockIdx.x;
  const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  int tile_m = wg / nt, tile_n = wg % nt;
  if (g.group_m > 1) {   // grouped order: the blocks an XCD runs concurrently cover group_m tile rows x few tile columns
    const int per = g.group_m * nt, first = (wg / per) * g.group_m, in = wg % per;
    const int gsz = mt - first < g.group_m ? mt - first : g.group_m;
    tile_m = first + in % gsz;
    tile_n = in / gsz;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // staging geometry: a plane tile is 128 rows x 4 uint4 (32 bf16); thread -> (row = tid>>2 (+64), part = tid&3)
  const int srow = tid >> 2, spart = tid & 3;
  int ma0 = m0 + srow, ma1 = m0 + srow + 64, nb0 = n0 + srow, nb1 = n0 + srow + 64;
  ma0 = ma0 < g.M ? ma0 : g.M - 1;   // clamp: rows beyond M/N are computed but never stored
  ma1 = ma1 < g.M ? ma1 : g.M - 1;
  nb0 = nb0 < g.N ? nb0 : g.N - 1;
  nb1 = nb1 < g.N ? nb1 : g.N - 1;
  const uint4* pa0h = reinterpret_cast<const uint4*>(g.A_hi + (size_t)ma0 * g.lda) + spart;
  const uint4* pa0l = reinterpret_cast<const uint4*>(g.A_lo + (size_t)ma0 * g.lda) + spart;
  const uint4* pa1h = reinterpret_cast<const uint4*>(g.A_hi + (size_t)ma1 * g.lda) + spart;
  const uint4* pa1l = reinterpret_cast<const uint4*>(g.A_lo + (size_t)ma1 * g.lda) + spart;
  const uint4* pb0h = reinterpret_cast<const uint4*>(g.W_hi + (size_t)nb0 * g.ldw) + spart;
  const uint4* pb0l = reinterpret_cast<const uint4*>(g.W_lo + (size_t)nb0 * g.ldw) + spart;
  const uint4* pb1h = reinterpret_cast<const uint4*>(g.W_hi + (size_t)nb1 * g.ldw) + spart;
  const uint4* pb1l = reinterpret_cast<const uint4*>(g.W_lo + (size_t)nb1 * g.ldw) + spart;
  const int so0 = srow * ROWB + spart * 16, so1 = (srow + 64) * ROWB + spart * 16;

  const int nk = g.Kp / BK;   // Kp (padded K) is a multiple of 32; pad columns are zero in both operands
  // Two named staging register sets => global loads run TWO K-tiles ahead of the MFMAs (the loop is latency-,
  // not bandwidth-bound with a single tile in flight).
  uint4 xa0h, xa0l, xa1h, xa1l, xb0h, xb0l, xb1h, xb1l;
  uint4 ya0h, ya0l, ya1h, ya1l, yb0h, yb0l, yb1h, yb1l;
#define G_LOAD(P, kt)                                                                     \
  {                                                                                       \
    const int ko = (kt) * 4;                                                              \
    P##a0h = pa0h[ko]; P##a0l = pa0l[ko]; P##a1h = pa1h[ko]; P##a1l = pa1l[ko];           \
    P##b0h = pb0h[ko]; P##b0l = pb0l[ko]; P##b1h = pb1h[ko]; P##b1l = pb1l[ko];           \
  }
#define G_STORE(P, buf)                                                                   \
  {                                                                                       \
    *reinterpret_cast<uint4*>(&lds[buf][0][0][so0]) = P##a0h;                             \
    *reinterpret_cast<uint4*>(&lds[buf][0][1][so0]) = P##a0l;                             \
    *reinterpret_cast<uint4*>(&lds[buf][0][0][so1]) = P##a1h;                             \
    *reinterpret_cast<uint4*>(&lds[buf][0][1][so1]) = P##a1l;                             \
    *reinterpret_cast<uint4*>(&lds[buf][1][0][so0]) = P##b0h;                             \
    *reinterpret_cast<uint4*>(&lds[buf][1][1][so0]) = P##b0l;                             \
    *reinterpret_cast<uint4*>(&lds[buf][1][0][so1]) = P##b1h;                             \
    *reinterpret_cast<uint4*>(&lds[buf][1][1][so1]) = P##b1l;                             \
  }
#define G_COMPUTE(cur)                                                                    \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                         \
    const int koff = s * 32 + half * 16;                                                  \
    bf16x8 fa0[2], fa1[2], fb0[2], fb1[2];                                                \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                       \
      const int ar = (wm * 64 + t * 32 + l31) * ROWB + koff;                              \
      const int br = (wn * 64 + t * 32 + l31) * ROWB + koff;                              \
      fa0[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][0][0][ar]);                     \
      fa1[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][0][1][ar]);                     \
      fb0[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][1][0][br]);                     \
      fb1[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][1][1][br]);                     \
    }                                                                                     \
    _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                      \
      _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                    \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[tm], fb0[tn], acc[tm][tn], 0, 0, 0); \
    _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                      \
      _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                    \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[tm], fb1[tn], acc[tm][tn], 0, 0, 0); \
    _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                      \
      _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                    \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[tm], fb0[tn], acc[tm][tn], 0, 0, 0); \
  }

#if DS2_GEMM_INTERLEAVE == 0
  // prologue: tile 0 -> LDS[0]; tile 1 -> set x.  All loads/stores in the steady state are UNCONDITIONAL (tile
  // index clamped to nk-1): a branch around a load makes the compiler's s_waitcnt placement conservative
  // (vmcnt(0) before the first ds_write), which serialises the two-deep prefetch.
  const int last = nk - 1;
  G_LOAD(x, 0)
  G_STORE(x, 0)
  G_LOAD(x, (1 < last ? 1 : last))
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {   // LDS[0] holds tile kt, set x holds tile kt+1
    if (!(DBG & 2)) G_LOAD(y, (kt + 2 < last ? kt + 2 : last))
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch issue ABOVE the MFMAs (the scheduler sinks it otherwise)
    if (!(DBG & 4)) G_COMPUTE(0)
    if (!(DBG & 1)) G_STORE(x, 1)
    __syncthreads();
    if (!(DBG & 2)) G_LOAD(x, (kt + 3 < last ? kt + 3 : last))
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4)) G_COMPUTE(1)
    if (!(DBG & 1)) G_STORE(y, 0)
    __syncthreads();
  }
  if (kt < nk) G_COMPUTE(0)   // odd tail: tile nk-1 sits in LDS[0]
#else
  // Interleaved steady state: the 8 ds_write_b128 of the NEXT tile and the 8 global loads of the tile after
  // next are issued in the shadow of the first 8 MFMAs of the current tile (an MFMA occupies the matrix pipe for 32
  // cycles; LDS / VMEM issue slots are free meanwhile), instead of after the 24th MFMA.
#define G_READ(cur, s, A0, A1, B0, B1)                                                    \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                         \
    const int ar = (wm * 64 + t * 32 + l31) * ROWB + (s) * 32 + half * 16;                \
    const int br = (wn * 64 + t * 32 + l31) * ROWB + (s) * 32 + half * 16;                \
    A0[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][0][0][ar]);                        \
    A1[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][0][1][ar]);                        \
    B0[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][1][0][br]);                        \
    B1[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][1][1][br]);                        \
  }
#define G_MFMA(A0, A1, B0, B1)                                                            \
  _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                        \
    _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                      \
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[tm], B0[tn], acc[tm][tn], 0, 0, 0); \
  _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                        \
    _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                      \
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[tm], B1[tn], acc[tm][tn], 0, 0, 0); \
  _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                        \
    _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                      \
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[tm], B0[tn], acc[tm][tn], 0, 0, 0);
#define G_ITER(cur, nxt, P, ktl)                                                          \
  {                                                                                       \
    bf16x8 fa0[2], fa1[2], fb0[2], fb1[2], ga0[2], ga1[2], gb0[2], gb1[2];                \
    G_READ(cur, 0, fa0, fa1, fb0, fb1)                                                    \
    G_MFMA(fa0, fa1, fb0, fb1)                                                            \
    G_STORE(P, nxt)                                                                       \
    G_LOAD(P, ktl)                                                                        \
    G_READ(cur, 1, ga0, ga1, gb0, gb1)                                                    \
    G_MFMA(ga0, ga1, gb0, gb1)                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                  \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                  \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                  \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                                   \
  }
  const int last = nk - 1;
  G_LOAD(x, 0)
  G_STORE(x, 0)
  G_LOAD(x, (1 < last ? 1 : last))
  G_LOAD(y, (2 < last ? 2 : last))
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {   // LDS[0] holds tile kt, set x tile kt+1, set y tile kt+2
    G_ITER(0, 1, x, (kt + 3 < last ? kt + 3 : last))
    __syncthreads();
    G_ITER(1, 0, y, (kt + 4 < last ? kt + 4 : last))
    __syncthreads();
  }
  if (kt < nk) G_COMPUTE(0)   // odd tail: tile nk-1 sits in LDS[0]
#endif

  // ---- epilogue, staged through LDS: the MFMA fragment layout (one column, 16 scattered rows per lane) would give
  // 4-byte scattered global stores; each wave parks its 64x64 tile in LDS and re-reads it row-wise so that every
  // lane handles 4 consecutive columns: 16-byte loads of bias/residual, 16-byte fp32 stores, 8-byte plane stores.
  __syncthreads();                                   // operand tiles are dead
  constexpr int EPLD = 68;                           // floats per staged row (64 + 4 pad)
  float* ep = reinterpret_cast<float*>(&lds[0][0][0][0]) + wave * (64 * EPLD);
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) ep[(tm * 32 + mfma32_row(e, half)) * EPLD + tn * 32 + l31] = acc[tm][tn][e];
  __syncthreads();
  if (DBG & 8) return;
  const int c4 = lane & 15, r0 = lane >> 4;
  const int n = n0 + wn * 64 + c4 * 4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), gam4 = make_float4(1.f, 1.f, 1.f, 1.f);
  {
    float* bp = reinterpret_cast<float*>(&bias4);
    float* gp = reinterpret_cast<float*>(&gam4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (g.bias && n + j < g.N) bp[j] = g.bias[n + j];
      if (g.gamma && n + j < g.N) gp[j] = g.gamma[n + j];
    }
  }
  const bool vec_ok = (n + 3 < g.N);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int r = it * 4 + r0;
    const int m = m0 + wm * 64 + r;
    if (m >= g.M) continue;
    const float4 a4 = *reinterpret_cast<const float4*>(&ep[r * EPLD + c4 * 4]);
    float v[4] = {a4.x + bias4.x, a4.y + bias4.y, a4.z + bias4.z, a4.w + bias4.w};
    ds2_act4(v, g.act);
    v[0] *= gam4.x; v[1] *= gam4.y; v[2] *= gam4.z; v[3] *= gam4.w;
    if (g.R) {
      const int rm = g.r_mod > 0 ? (m % g.r_mod) : m;
      const float* rp = g.R + (size_t)rm * g.ldr + n;
      if (vec_ok && (g.ldr & 3) == 0) {
        const float4 r4 = *reinterpret_cast<const float4*>(rp);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < g.N) v[j] += rp[j];
      }
    }
    if (g.C) {
      float* cp = g.C + (size_t)m * g.ldc + n;
      if (vec_ok && (g.ldc & 3) == 0) {
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < g.N) cp[j] = v[j];
      }
    }
    if (g.C_hi && n < g.ldcp) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j >= g.N) v[j] = 0.f;                 // pad columns of the planes are zero
      if (g.rope_cis) {   // columns (n, n+1), (n+2, n+3) are complex pairs (apply_rotary_enc, position_encoding.py:196-220)
        const int t = m % g.rope_L;
        if (t < g.rope_n) {
          const float4 c = *reinterpret_cast<const float4*>(g.rope_cis + ((size_t)(t % g.rope_grid) * 128 + (n >> 1)) * 2);
          const float a0 = v[0] * c.x - v[1] * c.y, a1 = v[0] * c.y + v[1] * c.x;
          const float a2 = v[2] * c.z - v[3] * c.w, a3 = v[2] * c.w + v[3] * c.z;
          v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
        }
      }
      uint2 h, l;
      h.x = cvt_pk_bf16(v[0], v[1]);
      h.y = cvt_pk_bf16(v[2], v[3]);
      l.x = cvt_pk_bf16(v[0] - bf_lo(h.x), v[1] - bf_hi(h.x));
      l.y = cvt_pk_bf16(v[2] - bf_lo(h.y), v[3] - bf_hi(h.y));
      *reinterpret_cast<uint2*>(g.C_hi + (size_t)m * g.ldcp + n) = h;
      if (g.C_lo) *reinterpret_cast<uint2*>(g.C_lo + (size_t)m * g.ldcp + n) = l;
    }
  }
}

}  // namespace

int launch_split_rows(const float* x, int ldx, int rows, int cols, void* hi, void* lo, int ldp, hipStream_t st, bool f16, int mx) {
  DS2_REQUIRE(ldp % 32 == 0 && ldx % 4 == 0 && cols % 4 == 0, "split_rows: ldp must be a multiple of 32, ldx/cols of 4");
  const size_t n = (size_t)rows * (ldp / 4);
  auto kern = mx == DS2_PLANES_MX_A ? k_split_rows_mx<false> : mx == DS2_PLANES_MX_W ? k_split_rows_mx<true> : f16 ? k_split_rows_f16 : k_split_rows;
  hipLaunchKernelGGL(kern, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ldx, rows, cols,
                     reinterpret_cast<uint2*>(hi), reinterpret_cast<uint2*>(lo), ldp);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_planes_bf16_to_mx(const void* hi, const void* lo, void* p1, void* p2, size_t elems, hipStream_t st) {
  DS2_REQUIRE(elems % 4 == 0, "planes_bf16_to_mx: element count must be a multiple of 4");
  const size_t n = elems / 4;
  hipLaunchKernelGGL(k_planes_bf16_to_mx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint2*>(hi),
                     reinterpret_cast<const uint2*>(lo), reinterpret_cast<uint2*>(p1), reinterpret_cast<uint2*>(p2), n);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

namespace {

// heuristic tile choice (cost model calibrated with tools/gemm_ablate.sh: cost = rounds * block_work / efficiency with
// efficiencies 1 : 1.15 : 1.28 for 128x128 (2 blocks/CU) : 256x128 three-stage ring : 256x256).  All variants are
// bound by the global->LDS operand path, so larger tiles win whenever they still fill the 256 CUs.
int heuristic_tile(const GemmSplitArgs& g) {
  const int ncols = g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N;
  const long b128 = (long)cdiv(g.M, 128) * cdiv(ncols, 128);
  const long br3 = (long)cdiv(g.M, 256) * cdiv(ncols, 128);
  const long b256 = (long)cdiv(g.M, 256) * cdiv(ncols, 256);
  const double c128 = 2.0 * (double)((b128 + 511) / 512);
  const double cr3 = br3 >= 256 ? 2.0 / 1.15 * (double)((br3 + 255) / 256) : 1e30;
  const double c256 = b256 >= 256 ? 4.0 / 1.28 * (double)((b256 + 255) / 256) : 1e30;
  int tile = 1;
  if (cr3 < c128 && cr3 <= c256) tile = 3;
  const bool fits32 = (size_t)g.M * g.lda * 2 < (1ull << 32) && (size_t)g.N * g.ldw * 2 < (1ull << 32);   // 32-bit DMA offsets
  if (c256 < c128 && c256 < cr3) tile = fits32 ? 5 : 3;   // the LDS-DMA 256x256 kernel (else the 256x128 ring)
  // persistent variant with loader / storer waves for GEMMs without residual / RoPE epilogue
  if (tile == 5 && gemm_split_pp256_supported(g)) tile = 10;
  // ... and it also beats the 256x128 ring wherever that one was chosen for its smaller column padding: the GEMMs of Hiera
  // stages 1-2 (K = 144 / 288: five to nine K tiles, i.e. mostly epilogue) run 10-25 % faster persistent
  // (profiles/r02at_tile_time_s12.txt)
  if (tile == 3 && gemm_split_pp256_supported(g) && b256 >= 256 && ncols > 128 && fits32) tile = 10;
  // the assembly kernel with the overlapped epilogue (gemm_x4g.hip) wherever it takes the shape and fills the chip
  // (DS2_GEMM_X4G=0: the kernels above, for A/B runs)
  const char* x4g_e = getenv("DS2_GEMM_X4G");   // (read per call: the tests compare both paths in one process)
  const bool x4g = !(x4g_e && atoi(x4g_e) == 0);
  if (x4g && (tile == 3 || tile == 5 || tile == 10)) {
    const int cfg = gemm_split_x4g_config(g, 0);
    if (cfg != 0 && (long)(g.M / (cfg == 42 ? 256 : 128)) * (g.N / (cfg == 42 ? 128 : 192)) >= gemm_x4g_ncu()) tile = 11;
  }
  return tile;
}

int launch_tile_impl(const GemmSplitArgs& g_in, int tile, hipStream_t st, const char** kname);
int launch_tile(const GemmSplitArgs& g, int tile, hipStream_t st) {
  if (!ds2_prof_kernels()) { const char* k; return launch_tile_impl(g, tile, st, &k); }
  // the kernel that will run is only known after the fall-backs: bracket with a provisional tag, then rename
  const char* k = "?";
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return launch_tile_impl(g, tile, st, &k);
  (void)hipEventRecord(a, st);
  const int rc = launch_tile_impl(g, tile, st, &k);
  (void)hipEventRecord(b, st);
  char tag[96];
  snprintf(tag, sizeof(tag), "kern %s %d %d %d", k, g.M, g.N, g.Kp);
  ds2_prof_record(tag, a, b);
  return rc;
}
int launch_tile_impl(const GemmSplitArgs& g_in, int tile, hipStream_t st, const char** kname) {
  GemmSplitArgs g = g_in;
  {   // tile order (see GemmSplitArgs::group_m): wide-N GEMMs get 8-row groups (measured again in round 5 on the assembly kernel:
      // 0 / 2 / 4 / 16 are all slower or equal, profiles/r05_d_x4g.txt; the DS2_GEMM_GROUPM / DS2_GEMM_PF overrides of round 2 are gone)
    const int bn = (tile == 5 || tile == 10) ? 256 : 128;
    const int ntl = cdiv(g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N, bn);
    g.group_m = ntl >= 8 ? 8 : 0;
    g.prefetch = 0;     // (k_gemm_split_d256's L2 touch loads: slower at every distance, GEMM findings 5)
  }
  // forced tiles (DS2_GEMM_TILE) fall back when a kernel cannot take the shape: persistent -> one-tile 256x256 -> ring
  const bool fits32 = (size_t)g.M * g.lda * 2 < (1ull << 32) && (size_t)g.N * g.ldw * 2 < (1ull << 32);   // 32-bit DMA offsets
  if (tile >= 11 && tile <= 13) {   // 11: assembly kernel, its own choice of configuration; 12: 256 x 128; 13: 128 x 192
    const int cfg = gemm_split_x4g_config(g, tile == 12 ? 42 : tile == 13 ? 23 : 0);
    if (cfg != 0) return launch_gemm_split_x4g(g, cfg, st, kname);
    tile = 10;
  }
  if (tile == 10 && !(gemm_split_pp256_supported(g) && fits32)) tile = 5;
  if (tile == 5 && !fits32) tile = 3;
  if (tile == 10) { *kname = "k_gemm_split_pp256"; return launch_gemm_split_pp256(g, st); }
  if (tile == 5) { *kname = "k_gemm_split_d256"; return launch_gemm_split_d256(g, st); }
  if (tile == 3) { *kname = "k_gemm_split_r3"; return launch_gemm_split_r3(g, st); }
  *kname = "k_gemm_split";
  const int mt = cdiv(g.M, BM), nt = cdiv(g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N, BN);
  // (the ablation instantiations k_gemm_split<1..15> of round 1 - wrong results by construction - are gone: tools/gemm_ablate*.sh
  // document what they measured)
  hipLaunchKernelGGL(k_gemm_split<0>, dim3(mt * nt), dim3(256), 0, st, g, mt, nt);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

}  // namespace

int launch_gemm_split(const GemmSplitArgs& g, hipStream_t st) {
  DS2_REQUIRE(g.M > 0 && g.N > 0 && g.Kp > 0 && g.Kp % 32 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 && g.lda >= g.Kp && g.ldw >= g.Kp,
              "gemm_split: bad dims M=%d N=%d Kp=%d lda=%d ldw=%d", g.M, g.N, g.Kp, g.lda, g.ldw);
  DS2_REQUIRE(g.C || g.C_hi, "gemm_split: no output");
  DS2_REQUIRE(!g.C_hi || (g.ldcp % 2 == 0), "gemm_split: ldcp must be even");
  if (g.mx) {   // MX planes: the assembly kernel's 128 x 192 configuration is the only consumer of that format
    DS2_REQUIRE(gemm_split_x4g_config(g, 23) == 23, "gemm_split: no MX kernel for M=%d N=%d Kp=%d (needs M %% 128 == 0, N %% 192 == 0, Kp %% 64 == 0, Kp >= 576)",
                g.M, g.N, g.Kp);
    GemmSplitArgs gm = g;
    gm.group_m = cdiv(g.N, 128) >= 8 ? 8 : 0;          // (the tile order of the bf16 form, launch_tile_impl)
    if (!ds2_prof_kernels()) { const char* k; return launch_gemm_split_x4g(gm, 23, st, &k); }
    const char* k = "?";
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return launch_gemm_split_x4g(gm, 23, st, &k);
    (void)hipEventRecord(a, st);
    const int rc = launch_gemm_split_x4g(gm, 23, st, &k);
    (void)hipEventRecord(b, st);
    char tag[96];
    snprintf(tag, sizeof(tag), "kern %s %d %d %d", k, g.M, g.N, g.Kp);
    ds2_prof_record(tag, a, b);
    return rc;
  }
  const char* tile_e = getenv("DS2_GEMM_TILE");   // (read per call: the tests force tiles in one process)
  const int tile_env = tile_e ? atoi(tile_e) : 0;
  if (g.c_hi_f16) {   // fp16 key planes (mode bf16x3k): the K = 64 streaming kernel's epilogue is the one that writes them
    DS2_REQUIRE(gemm_split_k64_supported(g), "gemm_split: fp16 hi planes are produced by the K = 64 kernel only (M=%d N=%d Kp=%d)", g.M, g.N, g.Kp);
  } else
  if (tile_env != 0) return launch_tile(g, tile_env, st);
  // K = 64 projections over hundreds of thousands of rows (memory-attention keys): HBM-bound weight-stationary kernel
  if (gemm_split_k64_supported(g)) {
    if (!ds2_prof_kernels()) return launch_gemm_split_k64(g, st);
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return launch_gemm_split_k64(g, st);
    (void)hipEventRecord(a, st);
    const int rc = launch_gemm_split_k64(g, st);
    (void)hipEventRecord(b, st);
    char tag[96];
    snprintf(tag, sizeof(tag), "kern k_gemm_split_k64 %d %d %d", g.M, g.N, g.Kp);
    ds2_prof_record(tag, a, b);
    return rc;
  }
  return launch_tile(g, heuristic_tile(g), st);
}
