// bf16x3 GEMM over pre-split operands, LARGE block tiles (256x256 / 256x128, 8 waves).
//
// Why: rocprofv3 PMC on the 128x128 kernel (gemm_split.hip) shows the L2 serving ~160 G requests/s of 64 bytes
// (TCC_REQ, 83 % hits) - the operand re-read rate, not the matrix pipe, bounds it: a bf16x3 operand element is 4 bytes
// (two planes), so a 128x128 tile has only 32 algorithmic FLOP per L2 byte.  A 256x256 tile halves the L2 bytes per
// FLOP (256x128: -25 %), and the longer K step (48 MFMAs per wave per 32-deep tile) hides the one-tile-ahead global
// prefetch behind a single staging register set.
//
// Block = 512 threads = 8 waves; wave tile = (32*MF) x 64:  MF = 4 -> waves 2 x 4, block 256 x 256
//                                                            MF = 2 -> waves 4 x 2, block 256 x 128
// LDS: 2 buffers x {A hi, A lo, W hi, W lo} x rows x 80 B (64 B of K + 16 B pad: conflict-free ds_read_b128).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BM2 = 256, BK = 32, ROWB = 80;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <int MF>
__global__ __launch_bounds__(512, 1) void k_gemm_split256(GemmSplitArgs g, int mt, int nt) {
  constexpr int BN2 = MF == 4 ? 256 : 128;
  constexpr int WN = BN2 / 64;                 // waves along n
  constexpr int PA = BM2 * ROWB, PB = BN2 * ROWB, BUF = 2 * PA + 2 * PB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
  // buffer b: A hi at b*BUF, A lo at +PA, W hi at +2PA, W lo at +2PA+PB

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
  const int m0 = tile_m * BM2, n0 = tile_n * BN2;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[MF][2];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // staging: thread -> (row = tid>>2 (+128), part = tid&3)
  const int srow = tid >> 2, spart = tid & 3;
  int ma0 = m0 + srow, ma1 = m0 + srow + 128, nb0 = n0 + srow, nb1 = n0 + srow + 128;
  ma0 = ma0 < g.M ? ma0 : g.M - 1;
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
  const int so0 = srow * ROWB + spart * 16, so1 = (srow + 128) * ROWB + spart * 16;

  const int nk = g.Kp / BK;
  const int last = nk - 1;
  uint4 xa0h, xa0l, xa1h, xa1l, xb0h, xb0l, xb1h, xb1l;
#define G2_LOAD(kt)                                                                       \
  {                                                                                       \
    const int ko = (kt) * 4;                                                              \
    xa0h = pa0h[ko]; xa0l = pa0l[ko]; xa1h = pa1h[ko]; xa1l = pa1l[ko];                   \
    xb0h = pb0h[ko]; xb0l = pb0l[ko];                                                     \
    if (MF == 4) { xb1h = pb1h[ko]; xb1l = pb1l[ko]; }                                    \
  }
#define G2_STORE(buf)                                                                     \
  {                                                                                       \
    unsigned char* b_ = lds + (buf) * BUF;                                                \
    *reinterpret_cast<uint4*>(b_ + so0) = xa0h;                                           \
    *reinterpret_cast<uint4*>(b_ + PA + so0) = xa0l;                                      \
    *reinterpret_cast<uint4*>(b_ + so1) = xa1h;                                           \
    *reinterpret_cast<uint4*>(b_ + PA + so1) = xa1l;                                      \
    *reinterpret_cast<uint4*>(b_ + 2 * PA + so0) = xb0h;                                  \
    *reinterpret_cast<uint4*>(b_ + 2 * PA + PB + so0) = xb0l;                             \
    if (MF == 4) {                                                                        \
      *reinterpret_cast<uint4*>(b_ + 2 * PA + so1) = xb1h;                                \
      *reinterpret_cast<uint4*>(b_ + 2 * PA + PB + so1) = xb1l;                           \
    }                                                                                     \
  }
#define G2_COMPUTE(buf)                                                                   \
  {                                                                                       \
    const unsigned char* b_ = lds + (buf) * BUF;                                          \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                       \
      const int koff = s * 32 + half * 16;                                                \
      bf16x8 fa0[MF], fa1[MF], fb0[2], fb1[2];                                            \
      _Pragma("unroll") for (int t = 0; t < MF; ++t) {                                    \
        const int ar = (wm * (MF * 32) + t * 32 + l31) * ROWB + koff;                     \
        fa0[t] = *reinterpret_cast<const bf16x8*>(b_ + ar);                               \
        fa1[t] = *reinterpret_cast<const bf16x8*>(b_ + PA + ar);                          \
      }                                                                                   \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                     \
        const int br = (wn * 64 + t * 32 + l31) * ROWB + koff;                            \
        fb0[t] = *reinterpret_cast<const bf16x8*>(b_ + 2 * PA + br);                      \
        fb1[t] = *reinterpret_cast<const bf16x8*>(b_ + 2 * PA + PB + br);                 \
      }                                                                                   \
      _Pragma("unroll") for (int tm = 0; tm < MF; ++tm)                                   \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                  \
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[tm], fb0[tn], acc[tm][tn], 0, 0, 0); \
      _Pragma("unroll") for (int tm = 0; tm < MF; ++tm)                                   \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                  \
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[tm], fb1[tn], acc[tm][tn], 0, 0, 0); \
      _Pragma("unroll") for (int tm = 0; tm < MF; ++tm)                                   \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                  \
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[tm], fb0[tn], acc[tm][tn], 0, 0, 0); \
    }                                                                                     \
  }

  // LDS[0] <- tile 0, registers <- tile 1.  Steady state (tile kt in LDS[kt&1], tile kt+1 in registers): park the
  // registers in the other buffer, refill them with tile kt+2 (a whole 48-MFMA step ahead of their use), compute.
  G2_LOAD(0)
  G2_STORE(0)
  G2_LOAD((1 < last ? 1 : last))
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    G2_STORE(1)
    G2_LOAD((kt + 2 < last ? kt + 2 : last))
    __builtin_amdgcn_sched_barrier(0);   // pin the prefetch ABOVE the MFMAs: hipcc sinks it to the end of the step otherwise
    G2_COMPUTE(0)
    __syncthreads();
    G2_STORE(0)
    G2_LOAD((kt + 3 < last ? kt + 3 : last))
    __builtin_amdgcn_sched_barrier(0);
    G2_COMPUTE(1)
    __syncthreads();
  }
  if (kt < nk) G2_COMPUTE(0)   // odd tail: tile nk-1 sits in LDS[0]

  // ---- epilogue: each wave parks one 32 x 64 slab of its tile in LDS at a time and re-reads it row-wise (4 consecutive
  // columns per lane: 16-byte bias/residual loads and fp32 stores, 8-byte plane stores); same arithmetic as
  // k_gemm_split's epilogue.
  __syncthreads();
  constexpr int EPLD = 68;
  float* ep = reinterpret_cast<float*>(lds) + wave * (32 * EPLD);
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
#pragma unroll
  for (int tm = 0; tm < MF; ++tm) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) ep[mfma32_row(e, half) * EPLD + tn * 32 + l31] = acc[tm][tn][e];
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int rr = it * 4 + r0;
      const int m = m0 + wm * (MF * 32) + tm * 32 + rr;
      if (m >= g.M) continue;
      const float4 a4 = *reinterpret_cast<const float4*>(&ep[rr * EPLD + c4 * 4]);
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
          if (n + j >= g.N) v[j] = 0.f;
        if (g.rope_cis) {   // apply_rotary_enc (position_encoding.py:196-220) on the complex pairs (n, n+1), (n+2, n+3)
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
    __syncthreads();
  }
}

}  // namespace

// mf = 4: 256x256 blocks, mf = 2: 256x128 blocks
int launch_gemm_split256(const GemmSplitArgs& g, int mf, hipStream_t st) {
  const int bn = mf == 4 ? 256 : 128;
  const int ncols = g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N;
  const int mt = cdiv(g.M, BM2), nt = cdiv(ncols, bn);
  if (mf == 4)
    hipLaunchKernelGGL(k_gemm_split256<4>, dim3(mt * nt), dim3(512), 0, st, g, mt, nt);
  else
    hipLaunchKernelGGL(k_gemm_split256<2>, dim3(mt * nt), dim3(512), 0, st, g, mt, nt);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
