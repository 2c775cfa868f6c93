// bf16x3 GEMM over pre-split operands: PERSISTENT 256x256-tile kernel with split memory roles (no residual / RoPE epilogue).
//
// Why (round 2): fitting T(K) of the 256x256 kernels at fixed M x N (tools/gemm_tile_time.sh, profiles/r02q_*) gives
// T = 169 us + 0.62 us * K for 65536 x 2304: a third of the fc1 GEMM (K = 576) is FIXED cost per output tile, ~19 us per
// 256x256 tile.  The store ablation (DS2_ABL_NOSTORE) attributes 7.7 us of it to the output itself: all 256 CUs finish
// their tiles together, write 64 MB together (8.5 TB/s, the fabric's limit) and a workgroup cannot retire - nor its
// successor start - before its stores have drained; the rest is workgroup dispatch, the exposed first DMA latency and the
// epilogue arithmetic.  A vector store only blocks a wave that WAITS on vmcnt, and on gfx950 loads and stores share that
// counter - so here
//  * one workgroup per CU loops over its output tiles (tile t -> workgroup t mod gridDim, the same XCD as the
//    one-tile-per-workgroup kernels' order);
//  * waves {0,1,4,5} ("loaders") issue every LDS-DMA piece and are the only ones that wait on vmcnt; waves {2,3,6,7}
//    ("storers") issue every global store of the epilogue and never wait on them: the stores of tile i drain under the
//    main loop of tile i+1;
//  * every wave applies bias / activation / gamma to its own accumulators in registers (so the GELU work stays spread over
//    all eight waves) and parks 32 x 64 slabs in LDS; a storer streams out its own slab and its loader partner's.
// Main loop, LDS image, hazards and per-element accumulation order: gemm_split_p256.hip (phase-interleaved, staggered
// wave groups); a loader issues the 4 pieces of rows [32 li, 32 li + 32) of a half-tile per phase and waits with vmcnt(16).
// Results are bit-identical to k_gemm_split_d256 / k_gemm_split_p256.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PBM = 256, PBN = 256, PBK = 32, PROWB = 64;
constexpr int HPL = 128 * PROWB;   // one plane of a half-tile: 8 KiB
constexpr int HT = 2 * HPL;        // half-tile (hi, lo): 16 KiB
constexpr int PSTAGE = 4 * HT;     // A0, A1, W0, W1: 64 KiB

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

struct FragW {
  bf16x8 h[2], l[2];   // [16-deep sub-step]
};

// DS2_PP_TRACE (profiling builds only): the waves of workgroup 0 stamp s_memtime around every barrier of their first tile
// (DS2_PP_TRACE=1: every step of the first tile; =2: the coarse events of the first three tiles, see tools/pp_trace.py)
#ifdef DS2_PP_TRACE
__device__ unsigned long long g_pp_trace[8][1024];
#define PP_STAMP()                                                                                 \
  if (trace_on && tix < 1024) {                                                                    \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                    \
    if (lane == 0) g_pp_trace[wave][tix] = t_;                                                     \
    ++tix;                                                                                         \
  }
#if DS2_PP_TRACE == 1
#define PP_T() PP_STAMP()
#define PP_TC()
#else
#define PP_T()
#define PP_TC() PP_STAMP()
#endif
#else
#define PP_T()
#define PP_TC()
#endif

__global__ __launch_bounds__(512, 1) void k_gemm_split_pp256(GemmSplitArgs g, int mt, int nt) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * PSTAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, half = lane >> 5;
  const bool loader = !(wave & 2);
  const int li = (wave & 1) | ((wave >> 2) << 1);   // loader index 0..3

  const char* bAh = reinterpret_cast<const char*>(g.A_hi);
  const char* bAl = reinterpret_cast<const char*>(g.A_lo);
  const char* bWh = reinterpret_cast<const char*>(g.W_hi);
  const char* bWl = reinterpret_cast<const char*>(g.W_lo);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nk = g.Kp / PBK;
  const int last = nk - 1;
  const int lc = (lane & 3) ^ ((lane >> 4) & 3);   // logical 16-byte chunk this lane fetches (XOR swizzle by row)
  const int sw = (l31 >> 2) & 3;
  const int fra = (wm * 64 + l31) * PROWB, frw = 2 * HT + (wn * 32 + l31) * PROWB;
  const int nwg = mt * nt;
  const int xq = nwg / 8, xr = nwg % 8;

#ifdef DS2_PP_TRACE
  int tix = 0;
#endif
  for (int orig = blockIdx.x; orig < nwg; orig += gridDim.x) {
#ifdef DS2_PP_TRACE
    const bool trace_on = DS2_PP_TRACE == 1 ? orig == 0 : (blockIdx.x == 0 && orig < 3 * (int)gridDim.x);
    PP_TC()   // 0: tile start
#endif
    const int xcd = orig % 8;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
    int tile_m = wg / nt, tile_n = wg % nt;
    if (g.group_m > 1) {
      const int per = g.group_m * nt, first = (wg / per) * g.group_m, in = wg % per;
      const int gsz = mt - first < g.group_m ? mt - first : g.group_m;
      tile_m = first + in % gsz;
      tile_n = in / gsz;
    }
    const int m0 = tile_m * PBM, n0 = tile_n * PBN;

    // epilogue constants of this wave's columns (tile tn: column n0 + tn*128 + wn*32 + l31); loaded now, used after the
    // main loop - for a storer that is also when the stores of its previous tile have long drained
    float bv[2], gv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int col = n0 + tn * 128 + wn * 32 + l31;
      bv[tn] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
      gv[tn] = (g.gamma && col < g.N) ? g.gamma[col] : 1.f;
    }

    // a wave's 32 columns of a 128-column half lie beyond N (N = 1152: the whole second half of the fifth tile): no MFMAs for them
    // (the wave still reads, stages and meets the barriers; on a power-limited chip the saved matrix work is time for the others)
    const bool deadq[2] = {n0 + wn * 32 >= g.N, n0 + 128 + wn * 32 >= g.N};
    f32x16 acc[4][2];   // [qm * 2 + 32-row tile][qn]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // DMA offsets: a half-tile is 16 pieces of 1 KiB (16 rows x 64 B of one plane); loader li issues rows [32 li, 32 li + 32)
    // of the hi and of the lo plane (4 pieces).  lane -> (row = lane >> 2, physical chunk = lane & 3).
    unsigned off[4][2];   // [A0, A1, W0, W1][16-row piece]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int ma = m0 + h * 128 + li * 32 + j * 16 + (lane >> 2);
        ma = ma < g.M ? ma : g.M - 1;   // clamp: rows beyond M/N are computed but never stored
        off[h][j] = ((unsigned)ma * (unsigned)g.lda + lc * 8) * 2u;
        int nb = n0 + h * 128 + li * 32 + j * 16 + (lane >> 2);
        nb = nb < g.N ? nb : g.N - 1;
        off[2 + h][j] = ((unsigned)nb * (unsigned)g.ldw + lc * 8) * 2u;
      }
#define PP_DMA(X, kt, so)                                                                                                 \
  if (loader) {                                                                                                           \
    const unsigned ko_ = (unsigned)((kt) < last ? (kt) : last) * (PBK * 2);                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                       \
      __builtin_amdgcn_global_load_lds(((X) < 2 ? bAh : bWh) + (off[X][j] + ko_),                                         \
                                       (lds_ptr)(lds + (so) + (X) * HT + (li * 2 + j) * 1024), 16, 0, 0);                 \
      __builtin_amdgcn_global_load_lds(((X) < 2 ? bAl : bWl) + (off[X][j] + ko_),                                         \
                                       (lds_ptr)(lds + (so) + (X) * HT + HPL + (li * 2 + j) * 1024), 16, 0, 0);           \
    }                                                                                                                     \
  }
    bf16x8 fah[2][2], fal[2][2];   // A fragments of the current A-half: [32-row tile][sub-step]
    FragW W0, W1;
#define PP_READ_A(h, so)                                                                                        \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                               \
    const unsigned char* b_ = lds + (so) + (h) * HT + fra + (((s * 2 + half) ^ sw) << 4);                       \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                             \
      fah[t][s] = *reinterpret_cast<const bf16x8*>(b_ + t * 32 * PROWB);                                        \
      fal[t][s] = *reinterpret_cast<const bf16x8*>(b_ + HPL + t * 32 * PROWB);                                  \
    }                                                                                                           \
  }
#define PP_READ_W(hh, so, F)                                                                                    \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                               \
    const unsigned char* b_ = lds + (so) + (hh) * HT + frw + (((s * 2 + half) ^ sw) << 4);                      \
    F.h[s] = *reinterpret_cast<const bf16x8*>(b_);                                                              \
    F.l[s] = *reinterpret_cast<const bf16x8*>(b_ + HPL);                                                        \
  }
#define PP_MFMA(qm, qn, F)                                                                                      \
  if (!deadq[qn])                                                                                               \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                               \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                             \
        acc[(qm) * 2 + t][qn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[t][s], F.h[s], acc[(qm) * 2 + t][qn], 0, 0, 0); \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                             \
        acc[(qm) * 2 + t][qn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[t][s], F.l[s], acc[(qm) * 2 + t][qn], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                               \
      acc[(qm) * 2 + t][qn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[t][s], F.h[s], acc[(qm) * 2 + t][qn], 0, 0, 0); \
  }
#define PP_WAITV()                                                                                              \
  __builtin_amdgcn_sched_barrier(0);                                                                            \
  if (loader) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                                 \
  __builtin_amdgcn_sched_barrier(0);
  // load segments of the four phases of K tile kt (fragment reads of the phase, one half-tile of DMA, counted wait)
#define PP_L0(kt) PP_READ_W(0, s0, W0) PP_READ_A(0, s0) __builtin_amdgcn_sched_barrier(0); PP_DMA(3, (kt) + 1, s1) PP_WAITV()
#define PP_L1(kt) PP_READ_W(1, s0, W1) __builtin_amdgcn_sched_barrier(0); PP_DMA(1, (kt) + 1, s1) PP_WAITV()
#define PP_L2(kt) PP_READ_A(1, s0) __builtin_amdgcn_sched_barrier(0); PP_DMA(0, (kt) + 2, s0) PP_WAITV()
#define PP_L3(kt) PP_DMA(2, (kt) + 2, s0) PP_WAITV()
  // matrix segment of one quadrant
#define PP_M(qm, qn, F, PRIO)                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                            \
  if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                      \
  PP_MFMA(qm, qn, F)                                                                                            \
  if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                      \
  __builtin_amdgcn_sched_barrier(0);
#define PP_BAR()                                                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                                            \
  PP_T()                                                                                                        \
  __builtin_amdgcn_s_barrier();                                                                                 \
  PP_T()                                                                                                        \
  __builtin_amdgcn_sched_barrier(0);

    // ONE barrier per phase ("step").  Waves 0-3 run a step as [load segment, matrix segment]; waves 4-7 run it as
    // [matrix segment of the PREVIOUS phase (at s_setprio 1), load segment]: on every SIMD the two co-resident waves are in
    // opposite segments without a second barrier per phase (an s_barrier costs ~140 cycles from last arrival to release:
    // tools/pp_trace.py).  Hazards in steps (a step ends with barrier k): a half-tile issued in step k is retired by every
    // loader's counted wait inside step k+4 and first read in step k+5; a slot read in step k (the lagging group retires
    // those reads at the head of step k+1) is re-issued in step k+2 or later.
    int s0 = 0, s1 = PSTAGE;
    PP_DMA(0, 0, s0) PP_DMA(2, 0, s0) PP_DMA(3, 0, s0) PP_DMA(1, 0, s0) PP_DMA(0, 1, s1) PP_DMA(2, 1, s1)
    PP_TC()   // 1: first DMA issued
    if (loader) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // A0[0], W0[0] landed ...
    __builtin_amdgcn_s_barrier();                                   // ... and are visible to everybody
    __builtin_amdgcn_sched_barrier(0);
    PP_TC()   // 2: first half-tiles landed
    if (wm == 0) {
      for (int kt = 0; kt < nk; ++kt) {
        PP_L0(kt) PP_T() PP_M(0, 0, W0, 0) PP_BAR()
        PP_L1(kt) PP_T() PP_M(0, 1, W1, 0) PP_BAR()
        PP_L2(kt) PP_T() PP_M(1, 1, W1, 0) PP_BAR()
        PP_L3(kt) PP_T() PP_M(1, 0, W0, 0) PP_BAR()
        const int t_ = s0; s0 = s1; s1 = t_;
      }
      PP_BAR()   // the lagging group's last matrix segment
    } else {
      PP_L0(0) PP_BAR()
      for (int kt = 0; kt < nk; ++kt) {
        PP_M(0, 0, W0, 1) PP_T() PP_L1(kt) PP_BAR()
        PP_M(0, 1, W1, 1) PP_T() PP_L2(kt) PP_BAR()
        PP_M(1, 1, W1, 1) PP_T() PP_L3(kt) PP_BAR()
        PP_M(1, 0, W0, 1) PP_T()
        const int t_ = s0; s0 = s1; s1 = t_;
        if (kt + 1 < nk) { PP_L0(kt + 1) }
        PP_BAR()
      }
    }

    // ---- epilogue.  Raw barriers + lgkmcnt only: a __syncthreads() would make the storers wait for their own stores.
    PP_TC()   // 3: main loop done
    if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (redundant) tail DMA must not land in the slabs
    __builtin_amdgcn_s_barrier();
    PP_TC()   // 4: tail DMA drained
    constexpr int EPLD = 68;
    float* ep_all = reinterpret_cast<float*>(lds);
    float* ep = ep_all + wave * (32 * EPLD);
    const int c4 = lane & 15, r0 = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {   // (fully unrolled: acc[] must be indexed statically)
      // (the activation is selected once per slab, not per element)
#define PP_PARK(ACT)                                                                                            \
  _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                                              \
    _Pragma("unroll") for (int e = 0; e < 16; ++e)                                                              \
      ep[mfma32_row(e, half) * EPLD + tn * 32 + l31] = ds2_act(acc[tm][tn][e] + bv[tn], ACT) * gv[tn];
      if (g.act == DS2_ACT_GELU) { PP_PARK(DS2_ACT_GELU) }
      else if (g.act == DS2_ACT_RELU) { PP_PARK(DS2_ACT_RELU) }
      else if (g.act == DS2_ACT_SIGMOID) { PP_PARK(DS2_ACT_SIGMOID) }
      else { PP_PARK(DS2_ACT_NONE) }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_TC()   // 5 + 3 tm: slab parked
      __builtin_amdgcn_s_barrier();
      if (!loader) {
#pragma unroll 1
        for (int sl = 0; sl < 2; ++sl) {
          const int ow = sl ? (wave ^ 2) : wave;   // slab owner: this wave, then its loader partner (same wm, wn - 2)
          const float* eps = ep_all + ow * (32 * EPLD);
          const int n = n0 + (c4 >> 3) * 128 + (ow & 3) * 32 + (c4 & 7) * 4;
          const bool vec_ok = (n + 3 < g.N);
#pragma unroll 4
          for (int it = 0; it < 8; ++it) {
            const int rr = it * 4 + r0;
            const int m = m0 + (tm >> 1) * 128 + wm * 64 + (tm & 1) * 32 + rr;
            if (m >= g.M) continue;
            const float4 a4 = *reinterpret_cast<const float4*>(&eps[rr * EPLD + c4 * 4]);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            if (g.C) {
              float* cp = g.C + (size_t)m * g.ldc + n;
              if (vec_ok && (g.ldc & 3) == 0) {   // (asm: hipcc otherwise merges this with the ragged path into dword + dwordx3)
                const f32x4 v4 = {v[0], v[1], v[2], v[3]};
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(cp), "v"(v4) : "memory");
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
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_TC()   // 6 + 3 tm: (storers) slabs streamed out
      __builtin_amdgcn_s_barrier();   // slabs free (for the next 32-row slab, or for the next tile's first DMA)
      PP_TC()   // 7 + 3 tm
    }
  }
}

}  // namespace

#ifdef DS2_PP_TRACE
extern "C" int ds2_debug_pp_trace(unsigned long long* out) {   // [8][1024] host buffer
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_trace), sizeof(unsigned long long) * 8 * 1024) == hipSuccess ? 0 : 1;
}
#endif

// (no residual: its loads would sit behind the storer's own stores in the in-order vmcnt; same for the RoPE table)
bool gemm_split_pp256_supported(const GemmSplitArgs& g) { return g.R == nullptr && g.rope_cis == nullptr; }

int launch_gemm_split_pp256(const GemmSplitArgs& g, hipStream_t st) {
  const int ncols = g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N;
  const int mt = cdiv(g.M, PBM), nt = cdiv(ncols, PBN);
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int grid = mt * nt < ncu ? mt * nt : ncu;
  hipLaunchKernelGGL(k_gemm_split_pp256, dim3(grid), dim3(512), 0, st, g, mt, nt);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
