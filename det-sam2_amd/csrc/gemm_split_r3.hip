// bf16x3 GEMM over pre-split operands: 256x128 block tile, 8 waves (wave tile 64x64), THREE-stage LDS-DMA ring.
//
// Ablation of the two-stage kernels (tools/gemm_ablate.sh): with the MFMAs removed, global->LDS staging + operand
// reads + barriers alone take as long as the full kernel - each K tile pays issue -> (loaded) memory latency
// (~2.5k cycles) -> barrier -> ds_read as a serial chain, because only ONE tile is in flight.  Here tile t+2 is
// issued before tile t+1 is waited for (counted s_waitcnt vmcnt(6): the 6 LDS-DMA pieces of the newest tile stay in
// flight across the raw s_barrier), so two tiles (96 KiB per CU) are always in flight.
// LDS: 3 stages x {A hi 16K, A lo 16K, W hi 8K, W lo 8K} = 144 KiB; rows are unpadded 64 B with the XOR swizzle
// chunk' = chunk ^ ((row >> 2) & 3) applied on the DMA source address and on the operand reads.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"


namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int RBM = 256, RBN = 128, BK = 32, ROWB = 64;
constexpr int PA = RBM * ROWB, PB = RBN * ROWB, STAGE = 2 * PA + 2 * PB;   // 49152

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

struct FragsR {
  bf16x8 ah[2], al[2], bh[2], bl[2];
};

__global__ __launch_bounds__(512, 1) void k_gemm_split_r3(GemmSplitArgs g, int mt, int nt) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * STAGE];

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
  const int m0 = tile_m * RBM, n0 = tile_n * RBN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // DMA sources: wave w stages A rows [32w, 32w+32) (2 pieces per plane) and W rows [16w, 16w+16) (1 piece per plane)
  const int lc = (lane & 3) ^ ((lane >> 4) & 3);               // logical 16-byte chunk this lane fetches
  unsigned oa[2], ob;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int ma = m0 + wave * 32 + 16 * j + (lane >> 2);
    ma = ma < g.M ? ma : g.M - 1;   // clamp: rows beyond M/N are computed but never stored
    oa[j] = ((unsigned)ma * (unsigned)g.lda + lc * 8) * 2u;
  }
  {
    int nb = n0 + wave * 16 + (lane >> 2);
    nb = nb < g.N ? nb : g.N - 1;
    ob = ((unsigned)nb * (unsigned)g.ldw + lc * 8) * 2u;
  }
  const char* bAh = reinterpret_cast<const char*>(g.A_hi);
  const char* bAl = reinterpret_cast<const char*>(g.A_lo);
  const char* bWh = reinterpret_cast<const char*>(g.W_hi);
  const char* bWl = reinterpret_cast<const char*>(g.W_lo);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nk = g.Kp / BK;
  const int last = nk - 1;
#define R3_DMA(src, dstoff) __builtin_amdgcn_global_load_lds((src), (lds_ptr)(lds + (dstoff)), 16, 0, 0);
#define R3_FILL(kt, so)                                                                        \
  {                                                                                            \
    const unsigned ko = (unsigned)((kt) < last ? (kt) : last) * (BK * 2);                      \
    R3_DMA(bAh + (oa[0] + ko), (so) + (wave * 32) * ROWB)                                      \
    R3_DMA(bAh + (oa[1] + ko), (so) + (wave * 32 + 16) * ROWB)                                 \
    R3_DMA(bAl + (oa[0] + ko), (so) + PA + (wave * 32) * ROWB)                                 \
    R3_DMA(bAl + (oa[1] + ko), (so) + PA + (wave * 32 + 16) * ROWB)                            \
    R3_DMA(bWh + (ob + ko), (so) + 2 * PA + (wave * 16) * ROWB)                                \
    R3_DMA(bWl + (ob + ko), (so) + 2 * PA + PB + (wave * 16) * ROWB)                           \
  }
  const int sw = (l31 >> 2) & 3;
  const int fra = (wm * 64 + l31) * ROWB, frb = 2 * PA + (wn * 64 + l31) * ROWB;
#define R3_READ(F, so, s)                                                                      \
  {                                                                                            \
    const unsigned char* b_ = lds + (so) + ((((s) * 2 + half) ^ sw) << 4);                     \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                            \
      F.ah[t] = *reinterpret_cast<const bf16x8*>(b_ + fra + t * 32 * ROWB);                    \
      F.al[t] = *reinterpret_cast<const bf16x8*>(b_ + PA + fra + t * 32 * ROWB);               \
      F.bh[t] = *reinterpret_cast<const bf16x8*>(b_ + frb + t * 32 * ROWB);                    \
      F.bl[t] = *reinterpret_cast<const bf16x8*>(b_ + PB + frb + t * 32 * ROWB);               \
    }                                                                                          \
  }
#define R3_MFMA_TERM(F, X, Y)                                                                  \
  _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                             \
    if (tn < live)                                                                             \
      _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                         \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.X[tm], F.Y[tn], acc[tm][tn], 0, 0, 0);

  // a wave whose columns lie (partly) beyond N (N = 576: the second wave column of the fifth 128-wide tile) multiplies zeros
  // there: it keeps staging, reading and meeting the barriers but issues no MFMA for its dead 32-column blocks - on a
  // power-limited chip that is time for the others.  live = 32-column blocks of this wave with at least one real column
  int live_ = (g.N - (n0 + wn * 64) + 31) / 32;
  live_ = live_ < 0 ? 0 : (live_ > 2 ? 2 : live_);
  const int live = __builtin_amdgcn_readfirstlane(live_);
  const bool dead = live == 0;
  FragsR F0, F1;
  int s0 = 0, s1 = STAGE, s2 = 2 * STAGE;       // stage offsets of tiles t, t+1, t+2
  R3_FILL(0, s0)
  R3_FILL(1, s1)
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();                      // ... and everybody else's
  R3_READ(F0, s0, 0)
  for (int kt = 0; kt < nk; ++kt) {
    // top of a step: tile kt in stage s0 (landed), its k-step-0 fragments in F0; tile kt+1 in flight into s1; s2 is
    // free (its last reader, tile kt-1's second k-step, was consumed before the previous step's barrier)
    R3_READ(F1, s0, 1)
    R3_FILL(kt + 2, s2)
    __builtin_amdgcn_sched_barrier(0);
    if (!dead) {
      R3_MFMA_TERM(F0, al, bh)
      R3_MFMA_TERM(F0, ah, bl)
      R3_MFMA_TERM(F0, ah, bh)
      R3_MFMA_TERM(F1, al, bh)
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile kt+1 landed; tile kt+2's 6 pieces stay in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    R3_READ(F0, s1, 0)
    __builtin_amdgcn_sched_barrier(0);   // keep the next tile's first reads AHEAD of the trailing MFMAs
    if (!dead) {
      R3_MFMA_TERM(F1, ah, bl)
      R3_MFMA_TERM(F1, ah, bh)
    }
    const int t_ = s0; s0 = s1; s1 = s2; s2 = t_;
  }

  // ---- epilogue (same as k_gemm_split): 32 x 64 slab per wave through LDS, row-wise 16-byte traffic
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (redundant) tail prefetches before LDS is reused
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
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) ep[mfma32_row(e, half) * EPLD + tn * 32 + l31] = acc[tm][tn][e];
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int rr = it * 4 + r0;
      const int m = m0 + wm * 64 + tm * 32 + rr;
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

int launch_gemm_split_r3(const GemmSplitArgs& g, hipStream_t st) {
  const int ncols = g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N;
  const int mt = cdiv(g.M, RBM), nt = cdiv(ncols, RBN);
  hipLaunchKernelGGL(k_gemm_split_r3, dim3(mt * nt), dim3(512), 0, st, g, mt, nt);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
