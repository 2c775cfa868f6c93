// bf16x3 GEMM over pre-split operands: 256x256 block tile, 8 waves (wave tile 128x64), TWO-stage LDS-DMA pipeline with
// double-buffered operand fragments and ONE barrier per 32-deep K tile.
//
// Why (round 2): tools/ubench/glds_path.hip measures 46-57 B/clk/CU through the L2 -> LDS path (global_load_lds and
// global_load_dwordx4 alike) while a 256x256x32 bf16x3 step needs only 21 B/clk/CU at full MFMA rate - the
// register-staged 256x256 kernel (gemm_split256.hip) is not operand-path bound, it is schedule bound: after every
// barrier all eight waves write their staging registers to LDS (64 KiB through ds_write_b128), then all read fragments,
// then all issue MFMAs, so the LDS phases and the matrix phase of the two waves of a SIMD never overlap.
// Here: no staging registers and no ds_write at all (LDS-DMA, source addresses pre-swizzled as in gemm_split_r3.hip);
// the fragments of K sub-step 1 are read while the 24 MFMAs of sub-step 0 run, and the fragments of the NEXT tile's
// sub-step 0 while the 24 MFMAs of sub-step 1 run; the single barrier of a tile sits between the two MFMA groups, and it
// both frees the stage just read (its refill for tile t+2 is issued right behind it and has a whole tile of MFMA time to
// land) and publishes tile t+1 (each wave waits for its own DMA pieces, issued one tile earlier, before arriving).
// LDS: 2 stages x {A hi, A lo, W hi, W lo} x 256 rows x 64 B = 128 KiB; XOR swizzle chunk' = chunk ^ ((row >> 2) & 3).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int DBM = 256, DBN = 256, BK = 32, ROWB = 64;
constexpr int PL = DBM * ROWB, STAGE = 4 * PL;   // 16 KiB per plane, 64 KiB per stage

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

struct FragsD {
  bf16x8 ah[4], al[4], bh[2], bl[2];
};

__global__ __launch_bounds__(512, 1) void k_gemm_split_d256(GemmSplitArgs g, int mt, int nt) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

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
  const int m0 = tile_m * DBM, n0 = tile_n * DBN;

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

  // Roles: waves 0-3 issue ALL LDS-DMA pieces of a tile (rows [64w, 64w+64) of each of the four planes: 16 pieces per
  // wave); waves 4-7 issue no DMA - they run the L2 prefetch instead (see below).  Both kinds of wave compute alike.
  const bool loader = wave < 4;
  const int lc = (lane & 3) ^ ((lane >> 4) & 3);               // logical 16-byte chunk this lane fetches
  unsigned oa[4], ob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int ma = m0 + (wave & 3) * 64 + 16 * j + (lane >> 2);
    ma = ma < g.M ? ma : g.M - 1;   // clamp: rows beyond M/N are computed but never stored
    oa[j] = ((unsigned)ma * (unsigned)g.lda + lc * 8) * 2u;
    int nb = n0 + (wave & 3) * 64 + 16 * j + (lane >> 2);
    nb = nb < g.N ? nb : g.N - 1;
    ob[j] = ((unsigned)nb * (unsigned)g.ldw + lc * 8) * 2u;
  }
  const char* bAh = reinterpret_cast<const char*>(g.A_hi);
  const char* bAl = reinterpret_cast<const char*>(g.A_lo);
  const char* bWh = reinterpret_cast<const char*>(g.W_hi);
  const char* bWl = reinterpret_cast<const char*>(g.W_lo);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nk = g.Kp / BK;
  const int last = nk - 1;
#define D2_DMA(src, dstoff) __builtin_amdgcn_global_load_lds((src), (lds_ptr)(lds + (dstoff)), 16, 0, 0);
#define D2_FILL(kt, so)                                                                        \
  if (loader) {                                                                                \
    const unsigned ko = (unsigned)((kt) < last ? (kt) : last) * (BK * 2);                      \
    const int ro = (wave * 64) * ROWB;                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
      D2_DMA(bAh + (oa[j] + ko), (so) + ro + j * 16 * ROWB)                                    \
      D2_DMA(bAl + (oa[j] + ko), (so) + PL + ro + j * 16 * ROWB)                               \
      D2_DMA(bWh + (ob[j] + ko), (so) + 2 * PL + ro + j * 16 * ROWB)                           \
      D2_DMA(bWl + (ob[j] + ko), (so) + 3 * PL + ro + j * 16 * ROWB)                           \
    }                                                                                          \
  }
  const int sw = (l31 >> 2) & 3;
  const int fra = (wm * 128 + l31) * ROWB, frb = 2 * PL + (wn * 64 + l31) * ROWB;
#define D2_READ(F, so, s)                                                                      \
  {                                                                                            \
    const unsigned char* b_ = lds + (so) + ((((s) * 2 + half) ^ sw) << 4);                     \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                            \
      F.bh[t] = *reinterpret_cast<const bf16x8*>(b_ + frb + t * 32 * ROWB);                    \
      F.bl[t] = *reinterpret_cast<const bf16x8*>(b_ + PL + frb + t * 32 * ROWB);               \
    }                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                            \
      F.ah[t] = *reinterpret_cast<const bf16x8*>(b_ + fra + t * 32 * ROWB);                    \
      F.al[t] = *reinterpret_cast<const bf16x8*>(b_ + PL + fra + t * 32 * ROWB);               \
    }                                                                                          \
  }
  // same per-element accumulation order as the other bf16x3 GEMM kernels: per 16-deep sub-step lo*hi, hi*lo, hi*hi
#define D2_MFMA_TERM(F, X, Y)                                                                  \
  _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                             \
    _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                           \
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.X[tm], F.Y[tn], acc[tm][tn], 0, 0, 0);

  // L2 prefetch.  A K tile is published by ONE barrier, so its latency is that of its slowest cache line - and some
  // lines always miss L2 (activations come from HBM / Infinity Cache on first touch; a weight matrix of several MiB
  // does not stay in the 4 MiB L2 while the activations stream through).  PMC: the waves spend ~40 % of their cycles in
  // s_waitcnt / s_barrier.  Waves 4-7 therefore "touch" one dword of every 64-byte row segment of K tile kt + g.prefetch
  // (64 rows x {A hi, A lo, W hi, W lo} per wave = 4 loads per step) so that the DMA of that tile, issued g.prefetch
  // steps later by waves 0-3, hits in L2.  The touches are fire-and-forget: these waves never wait on vmcnt, and the
  // loaders' vmcnt only counts their own DMA pieces (the memory counter is in-order per wave, which is why the touches
  // cannot ride in the loaders).  `junk` is read-write in every asm so that its register stays reserved from the first
  // touch to the end of the loop: a dead destination register would be re-allocated while loads into it are in flight.
  const int pf = g.prefetch;
  unsigned ta, tw;
  {
    int mr = m0 + (wave & 3) * 64 + lane;
    mr = mr < g.M ? mr : g.M - 1;
    ta = (unsigned)mr * (unsigned)g.lda * 2u;
    int nr = n0 + (wave & 3) * 64 + lane;
    nr = nr < g.N ? nr : g.N - 1;
    tw = (unsigned)nr * (unsigned)g.ldw * 2u;
  }
  unsigned junk = 0;
#define D2_TOUCH(kt)                                                                           \
  if (!loader && pf > 0 && (kt) + pf <= last) {                                                \
    const unsigned ko_ = (unsigned)((kt) + pf) * (BK * 2);                                     \
    const char *p0_ = bAh + (ta + ko_), *p1_ = bAl + (ta + ko_), *p2_ = bWh + (tw + ko_), *p3_ = bWl + (tw + ko_); \
    asm volatile("global_load_dword %0, %1, off\n\tglobal_load_dword %0, %2, off\n\t"          \
                 "global_load_dword %0, %3, off\n\tglobal_load_dword %0, %4, off"              \
                 : "+v"(junk) : "v"(p0_), "v"(p1_), "v"(p2_), "v"(p3_) : "memory");            \
  }

  FragsD F0, F1;
  int s0 = 0, s1 = STAGE;       // stage offsets of tiles t, t+1
  for (int t = 2 - pf; t < 2; ++t) D2_TOUCH(t)          // tiles 2 .. pf+1 (tiles 0 and 1 are fetched right away)
  D2_FILL(0, s0)
  D2_FILL(1, s1)
  if (loader) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile 0 landed (this wave's pieces) ...
  __builtin_amdgcn_s_barrier();                      // ... and everybody else's
  D2_READ(F0, s0, 0)
  for (int kt = 0; kt < nk; ++kt) {
    // top of a step: tile kt in stage s0 (landed, published), its sub-step-0 fragments in F0 (reads may still be in
    // flight); tile kt+1 in flight into s1 (issued one step ago)
    // (the first MFMA term goes ahead of the sub-step-1 reads: hipcc guards the first use of F0 with lgkmcnt(0), which
    // would otherwise also wait for the twelve reads just issued)
    D2_MFMA_TERM(F0, al, bh)
    __builtin_amdgcn_sched_barrier(0);
    D2_READ(F1, s0, 1)
    __builtin_amdgcn_sched_barrier(0);
    D2_MFMA_TERM(F0, ah, bl)
    D2_MFMA_TERM(F0, ah, bh)
    __builtin_amdgcn_sched_barrier(0);
    // F1 landed => this wave no longer reads stage s0; its own pieces of tile kt+1 landed.  After the barrier: stage s0
    // is free for tile kt+2 and tile kt+1 is visible to everyone.
    if (loader) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    D2_FILL(kt + 2, s0)
    D2_TOUCH(kt + 2)
    __builtin_amdgcn_sched_barrier(0);
    D2_MFMA_TERM(F1, al, bh)
    __builtin_amdgcn_sched_barrier(0);
    D2_READ(F0, s1, 0)
    __builtin_amdgcn_sched_barrier(0);
    D2_MFMA_TERM(F1, ah, bl)
    D2_MFMA_TERM(F1, ah, bh)
    const int t_ = s0; s0 = s1; s1 = t_;
  }
  asm volatile("" ::"v"(junk));

  // ---- epilogue: each wave parks one 32 x 64 slab of its tile in LDS at a time and re-reads it row-wise (4 consecutive
  // columns per lane: 16-byte bias/residual loads and fp32 stores, 8-byte plane stores); same arithmetic as
  // k_gemm_split's epilogue.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (redundant) tail DMA / touches before LDS is reused
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

int launch_gemm_split_d256(const GemmSplitArgs& g, hipStream_t st) {
  const int ncols = g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N;
  const int mt = cdiv(g.M, DBM), nt = cdiv(ncols, DBN);
  hipLaunchKernelGGL(k_gemm_split_d256, dim3(mt * nt), dim3(512), 0, st, g, mt, nt);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
