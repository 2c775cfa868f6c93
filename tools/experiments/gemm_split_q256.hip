// bf16x3 GEMM over pre-split operands: 256x256 block tile, 8 waves (wave tile 128x64), FOUR-stage LDS-DMA ring of 16-deep
// K slices with THREE slices in flight.
//
// Why: the ablation of the two-stage kernel (gemm_split_d256.hip; DESIGN.md "GEMM findings") shows that its DMA skeleton
// alone needs ~4k cycles per 32-deep tile - one 64 KiB tile in flight per CU, published by a barrier, costs the latency of
// its slowest cache line - against 3.1k cycles of MFMA work, and that the two overlap only partly.  A third 64 KiB stage
// does not fit the 160 KiB LDS.  Slicing K finer does: a 16-deep slice of the four planes is 32 KiB, four stages are
// 128 KiB, and a slice issued at the top of step t is needed only at the end of step t+2 - three steps (3 x 1536 SIMD
// cycles) of latency tolerance with 96 KiB in flight instead of one step with 64 KiB.  The price is a barrier per 24
// MFMAs of a wave instead of per 48.
// LDS image of a slice and plane: 256 rows x 32 B, rows consecutive - a wave's ds_read_b128 fragment (row = lane & 31,
// 16-byte half = lane >> 5) covers 1 KiB contiguously, conflict-free without a swizzle; a DMA piece (1 KiB per wave
// instruction) is 32 rows x 32 B, lane -> (row = lane >> 1, half = lane & 1).
// Same per-element accumulation order as the other bf16x3 GEMM kernels (per 16-deep sub-step lo*hi, hi*lo, hi*hi).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int QBM = 256, QBN = 256, QK = 16, QROWB = 32;
constexpr int QPL = QBM * QROWB, QSTAGE = 4 * QPL, QNST = 4;   // 8 KiB per plane, 32 KiB per stage

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

struct FragsQ {
  bf16x8 ah[4], al[4], bh[2], bl[2];
};

__global__ __launch_bounds__(512, 1) void k_gemm_split_q256(GemmSplitArgs g, int mt, int nt) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[QNST * QSTAGE];

  const int nwg = mt * nt;
  const int orig = blockIdx.x;
  const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  int tile_m = wg / nt, tile_n = wg % nt;
  if (g.group_m > 1) {
    const int per = g.group_m * nt, first = (wg / per) * g.group_m, in = wg % per;
    const int gsz = mt - first < g.group_m ? mt - first : g.group_m;
    tile_m = first + in % gsz;
    tile_n = in / gsz;
  }
  const int m0 = tile_m * QBM, n0 = tile_n * QBN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // DMA: wave w stages rows [32w, 32w+32) of each of the four planes - one 1 KiB piece per plane and slice
  unsigned oa, ob;
  {
    int ma = m0 + wave * 32 + (lane >> 1);
    ma = ma < g.M ? ma : g.M - 1;   // clamp: rows beyond M/N are computed but never stored
    oa = ((unsigned)ma * (unsigned)g.lda + (lane & 1) * 8) * 2u;
    int nb = n0 + wave * 32 + (lane >> 1);
    nb = nb < g.N ? nb : g.N - 1;
    ob = ((unsigned)nb * (unsigned)g.ldw + (lane & 1) * 8) * 2u;
  }
  const char* bAh = reinterpret_cast<const char*>(g.A_hi);
  const char* bAl = reinterpret_cast<const char*>(g.A_lo);
  const char* bWh = reinterpret_cast<const char*>(g.W_hi);
  const char* bWl = reinterpret_cast<const char*>(g.W_lo);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nk = g.Kp / QK;          // Kp is a multiple of 32
  const int last = nk - 1;
#define Q2_DMA(src, dstoff) __builtin_amdgcn_global_load_lds((src), (lds_ptr)(lds + (dstoff)), 16, 0, 0);
#define Q2_FILL(kt)                                                                            \
  {                                                                                            \
    const int kk_ = (kt) < last ? (kt) : last;                                                 \
    const unsigned ko = (unsigned)kk_ * (QK * 2);                                              \
    const int so = ((kt) & (QNST - 1)) * QSTAGE + (wave * 32) * QROWB;                         \
    Q2_DMA(bAh + (oa + ko), so)                                                                \
    Q2_DMA(bAl + (oa + ko), so + QPL)                                                          \
    Q2_DMA(bWh + (ob + ko), so + 2 * QPL)                                                      \
    Q2_DMA(bWl + (ob + ko), so + 3 * QPL)                                                      \
  }
  const int fra = (wm * 128 + l31) * QROWB + half * 16, frb = 2 * QPL + (wn * 64 + l31) * QROWB + half * 16;
#define Q2_READ(F, kt)                                                                         \
  {                                                                                            \
    const unsigned char* b_ = lds + ((kt) & (QNST - 1)) * QSTAGE;                              \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                            \
      F.bh[t] = *reinterpret_cast<const bf16x8*>(b_ + frb + t * 32 * QROWB);                   \
      F.bl[t] = *reinterpret_cast<const bf16x8*>(b_ + QPL + frb + t * 32 * QROWB);             \
    }                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                            \
      F.ah[t] = *reinterpret_cast<const bf16x8*>(b_ + fra + t * 32 * QROWB);                   \
      F.al[t] = *reinterpret_cast<const bf16x8*>(b_ + QPL + fra + t * 32 * QROWB);             \
    }                                                                                          \
  }
#define Q2_MFMA_TERM(F, X, Y)                                                                  \
  _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                             \
    _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                           \
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.X[tm], F.Y[tn], acc[tm][tn], 0, 0, 0);
  // one step = one 16-deep slice.  Top of step t: slices <= t+1 published, fragments of slice t in FA (reads possibly in
  // flight), slices t+2 and t+3 in flight, stage t%4 free (slice t lives in registers) -> refill it with slice t+4.
#define Q2_STEP(FA, FB, t)                                                                     \
  {                                                                                            \
    Q2_FILL((t) + 4)                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (!(DS2_EXP_GEMM2A || (g.drop_terms & 1))) Q2_MFMA_TERM(FA, al, bh)                                              \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    Q2_READ(FB, (t) + 1)                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (!(DS2_EXP_GEMM2W || (g.drop_terms & 2))) Q2_MFMA_TERM(FA, ah, bl)                                              \
    Q2_MFMA_TERM(FA, ah, bh)                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    /* slice t+2 landed (this wave's pieces; the 8 pieces of t+3, t+4 may fly on), FB landed */\
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                \
    __builtin_amdgcn_s_barrier();                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }

  FragsQ F0, F1;
  Q2_FILL(0)
  Q2_FILL(1)
  Q2_FILL(2)
  Q2_FILL(3)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // slices 0 and 1 landed (this wave's pieces) ...
  __builtin_amdgcn_s_barrier();                      // ... and everybody else's
  Q2_READ(F0, 0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // stage 0 is free from here on (slice 0 sits in registers)
  for (int kt = 0; kt < nk; kt += 2) {
    Q2_STEP(F0, F1, kt)
    Q2_STEP(F1, F0, kt + 1)
  }

  // ---- epilogue: each wave parks one 32 x 64 slab of its tile in LDS at a time and re-reads it row-wise (4 consecutive
  // columns per lane: 16-byte bias/residual loads and fp32 stores, 8-byte plane stores); same arithmetic as
  // k_gemm_split's epilogue.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (redundant) tail DMA before LDS is reused
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
  for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) ep[mfma32_row(e, half) * EPLD + tn * 32 + l31] = acc[tm][tn][e];
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int rr = it * 4 + r0;
      const int m = m0 + wm * (4 * 32) + tm * 32 + rr;
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

int launch_gemm_split_q256(const GemmSplitArgs& g, hipStream_t st) {
  const int ncols = g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N;
  const int mt = cdiv(g.M, QBM), nt = cdiv(ncols, QBN);
  hipLaunchKernelGGL(k_gemm_split_q256, dim3(mt * nt), dim3(512), 0, st, g, mt, nt);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
