// bf16x3 GEMM over pre-split operands for K = 64 and N <= 256: WEIGHT-STATIONARY, HBM-bound by design.
//
// The memory attention's key projection is C[B*Nk, 256] = kin[B*Nk, 64] . Wk^T (+ RoPE, emitted as key planes): at 16
// objects and a 7-frame bank M = 459 776 rows, K = 64.  A tile kernel pays a prologue, two K tiles and a 256 x 256 epilogue
// per block and reaches 50-60 TFLOP/s; but the GEMM is a stream: 118 MB of operand planes in, 235-471 MB of key planes
// out, 0.1 GFLOP per MB - its floor is the HBM time (~80-110 us), not the matrix pipe.  So:
//   * the whole weight (N x 64, both planes: <= 64 KiB) is loaded into LDS ONCE per workgroup (XOR-swizzled rows);
//   * a wave owns 32 rows x all N columns at a time: its A fragments come straight from global memory in MFMA fragment
//     shape (8 x 16-byte loads per lane, issued for the next tile ahead of this tile's last epilogue slab) - an A element is used by this wave only, there is
//     nothing to share through LDS;
//   * the workgroups are persistent (grid ~ 2 x CUs x ...) and stride over the row tiles; the epilogue (the same LDS-staged,
//     row-wise one as the tile kernels: bias, act, gamma, residual, RoPE, plane / fp32 output) streams the result out.
// Same per-element accumulation order as the tile kernels (per 16-deep sub-step: lo*hi, hi*lo, hi*hi), so results are
// bit-identical to them.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int KW = 64, WROWB = 128;   // K, bytes per weight row and plane in LDS

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// DS2_K64_TRACE (profiling builds only): waves of workgroup 0 stamp s_memtime along their first tiles (tools/k64_trace.py)
#ifdef DS2_K64_TRACE
__device__ unsigned long long g_k64_trace[8][256];
#define K64_T()                                                                                    \
  if (blockIdx.x == 0 && tix < 256 && (DS2_K64_TRACE == 1 || (g.rope_cis && g.M > 400000))) {                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                    \
    if (lane == 0) g_k64_trace[wave][tix] = t_;                                                    \
    ++tix;                                                                                         \
  }
#else
#define K64_T()
#endif

struct AFrags {
  bf16x8 h[4], l[4];   // four 16-deep sub-steps
};

// NT = N / 32 column tiles (4 or 8).  PLANES = the launcher found the key / value projection's epilogue (no activation, gamma,
// residual or fp32 output; plane output): those operands become compile-time constants and their (wave-uniform) branches
// disappear from the row loop, which is instruction-issue bound.
// F16H (PLANES only; GemmSplitArgs::c_hi_f16): the hi plane is written as IEEE fp16 and there is no lo plane - the single-plane
// keys of the memory attention in mode bf16x3k.
template <int NT, bool PLANES, bool F16H = false>
__global__ __launch_bounds__(512, 1) void k_gemm_split_k64(GemmSplitArgs g, int tiles_per_wave) {
  constexpr int NCOL = NT * 32;
  constexpr int WPL = NCOL * WROWB;                       // bytes of one weight plane
  constexpr int EPLD = 68;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WPL + 8 * 32 * EPLD * 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int e_act = PLANES ? (int)DS2_ACT_NONE : g.act;
  const float* const e_gamma = PLANES ? nullptr : g.gamma;
  const float* const e_R = PLANES ? nullptr : g.R;
  float* const e_C = PLANES ? nullptr : g.C;
  unsigned short* const e_Chi = g.C_hi;

  // ---- weights -> LDS once: row n, plane p, 16-byte chunk c at  p*WPL + n*128 + ((c ^ ((n >> 1) & 7)) << 4)
  for (int i = tid; i < NCOL * 8 * 2; i += 512) {
    const int p = i / (NCOL * 8), r = (i / 8) % NCOL, c = i & 7;
    const int nr = r < g.N ? r : g.N - 1;
    const uint4 v = *reinterpret_cast<const uint4*>((p ? g.W_lo : g.W_hi) + (size_t)nr * g.ldw + c * 8);
    *reinterpret_cast<uint4*>(lds + p * WPL + r * WROWB + ((c ^ ((r >> 1) & 7)) << 4)) = v;
  }
  __syncthreads();

  float* ep = reinterpret_cast<float*>(lds + 2 * WPL) + wave * (32 * EPLD);
  const int c4 = lane & 15, r0 = lane >> 4;
  const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
  const int ntiles = (g.M + 31) / 32;

  auto load_a = [&](int tile, AFrags& F) {
    int row = tile * 32 + l31;
    row = row < g.M ? row : g.M - 1;
    const unsigned short* ph = g.A_hi + (size_t)row * g.lda + half * 8;
    const unsigned short* pl = g.A_lo + (size_t)row * g.lda + half * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      F.h[s] = *reinterpret_cast<const bf16x8*>(ph + s * 16);
      F.l[s] = *reinterpret_cast<const bf16x8*>(pl + s * 16);
    }
  };

  AFrags F;
  {
    const int t0 = gw < ntiles ? gw : ntiles - 1;
    load_a(t0, F);
  }
  const int wsw = (l31 >> 1) & 7;
#ifdef DS2_K64_TRACE
  int tix = 0;
#endif
  for (int it = 0; it < tiles_per_wave; ++it) {
    K64_T()   // 0: tile start
    const int tile_raw = gw + it * nw;
    const int tile = tile_raw < ntiles ? tile_raw : ntiles - 1;      // surplus iterations recompute the last tile, store nothing
    const bool live = tile_raw < ntiles;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int coff = (((s * 2 + half) ^ wsw) << 4);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned char* wp = lds + (t * 32 + l31) * WROWB + coff;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wp);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wp + WPL);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.l[s], bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.h[s], bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.h[s], bh, acc[t], 0, 0, 0);
      }
    }
    K64_T()   // 1: MFMAs issued
    // ---- epilogue: 64-column slabs through the wave's LDS area, row-wise 16-byte traffic (as the tile kernels)
    const int m0 = tile * 32;
    // RoPE: the table rows of this lane's 8 output rows are the same for every slab; their (L2 / Infinity-Cache) loads are
    // issued together ahead of the slab's LDS round trip - one dependent load per row cost ~1500 cycles per row
    // (tools/k64_trace_bench.py: 75 % of the key projection)
    int rope_t[8];
    if (e_Chi && g.rope_cis) {
#pragma unroll
      for (int i8 = 0; i8 < 8; ++i8) {
        int m = m0 + i8 * 4 + r0;
        m = m < g.M ? m : g.M - 1;
        const int t = m % g.rope_L;
        rope_t[i8] = t < g.rope_n ? (t % g.rope_grid) : -1;   // (-1: object-pointer tokens are not rotated)
      }
    }
    {
#pragma unroll
    for (int sl = 0; sl < NT / 2; ++sl) {
      const int n = sl * 64 + c4 * 4;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), gam4 = make_float4(1.f, 1.f, 1.f, 1.f);
      {
        float* bp = reinterpret_cast<float*>(&bias4);
        float* gp = reinterpret_cast<float*>(&gam4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (g.bias && n + j < g.N) bp[j] = g.bias[n + j];
          if (e_gamma && n + j < g.N) gp[j] = e_gamma[n + j];
        }
      }
      const bool vec_ok = (n + 3 < g.N);
      if (sl == NT / 2 - 1) {   // the next row tile's fragments: fetched when most accumulators are dead (register budget),
        const int tn_raw = gw + (it + 1) * nw;   // they land under the last slab's LDS round trip and stores
        load_a(tn_raw < ntiles ? tn_raw : ntiles - 1, F);
      }
      __builtin_amdgcn_sched_barrier(0);   // (keeps the next slabs' table loads from being hoisted up here: register budget)
      float4 cis[8];
      if (e_Chi && g.rope_cis && n < g.ldcp) {
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8)
        {
          int pos = rope_t[i8] < 0 ? 0 : rope_t[i8];
          if (g.rope_w > 0) pos = (n >> 1) < 64 ? pos % g.rope_w : pos - pos % g.rope_w;   // compact rows, see GemmSplitArgs::rope_w
          cis[i8] = *reinterpret_cast<const float4*>(g.rope_cis + ((size_t)pos * 128 + (n >> 1)) * 2);
        }
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) ep[mfma32_row(e, half) * EPLD + tn * 32 + l31] = acc[sl * 2 + tn][e];
      // the slab is private to this wave and a wave's LDS operations complete in order: no workgroup barrier, the waves
      // run their tiles independently (only the compiler must not move the reads above the writes)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      K64_T()   // 2 + 2 sl: slab parked
#pragma unroll
      for (int i8 = 0; i8 < 8; ++i8) {
        const int rr = i8 * 4 + r0;
        const int m = m0 + rr;
        if (!live || m >= g.M) continue;
        const float4 a4 = *reinterpret_cast<const float4*>(&ep[rr * EPLD + c4 * 4]);
        float v[4] = {a4.x + bias4.x, a4.y + bias4.y, a4.z + bias4.z, a4.w + bias4.w};
        ds2_act4(v, e_act);
        v[0] *= gam4.x; v[1] *= gam4.y; v[2] *= gam4.z; v[3] *= gam4.w;
        if (e_R) {
          const int rm = g.r_mod > 0 ? (m % g.r_mod) : m;
          const float* rp = e_R + (size_t)rm * g.ldr + n;
          if (vec_ok && (g.ldr & 3) == 0) {
            const float4 r4 = *reinterpret_cast<const float4*>(rp);
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (n + j < g.N) v[j] += rp[j];
          }
        }
        if (e_C) {
          float* cp = e_C + (size_t)m * g.ldc + n;
          if (vec_ok && (g.ldc & 3) == 0) {
            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (n + j < g.N) cp[j] = v[j];
          }
        }
        if (e_Chi && n < g.ldcp) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j >= g.N) v[j] = 0.f;
          if (g.rope_cis) {   // apply_rotary_enc (position_encoding.py:196-220) on the complex pairs (n, n+1), (n+2, n+3)
            if (rope_t[i8] >= 0) {
              const float4 c = cis[i8];
              const float a0 = v[0] * c.x - v[1] * c.y, a1 = v[0] * c.y + v[1] * c.x;
              const float a2 = v[2] * c.z - v[3] * c.w, a3 = v[2] * c.w + v[3] * c.z;
              v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
            }
          }
          if constexpr (F16H) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            uint2 h;   // v_cvt_pk_f16_f32, round to nearest even
            h.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{ds2_sat_f16(v[0]), ds2_sat_f16(v[1])}), f16x2));
            h.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{ds2_sat_f16(v[2]), ds2_sat_f16(v[3])}), f16x2));
            *reinterpret_cast<uint2*>(e_Chi + (size_t)m * g.ldcp + n) = h;
          } else {
          uint2 h, l;
          h.x = cvt_pk_bf16(v[0], v[1]);
          h.y = cvt_pk_bf16(v[2], v[3]);
          l.x = cvt_pk_bf16(v[0] - bf_lo(h.x), v[1] - bf_hi(h.x));
          l.y = cvt_pk_bf16(v[2] - bf_lo(h.y), v[3] - bf_hi(h.y));
          *reinterpret_cast<uint2*>(e_Chi + (size_t)m * g.ldcp + n) = h;
          if (g.C_lo) *reinterpret_cast<uint2*>(g.C_lo + (size_t)m * g.ldcp + n) = l;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      K64_T()   // 3 + 2 sl: slab streamed out
    }
    }
  }
}


// ---- round 5: the key projection's epilogue WITHOUT the LDS round trip and WITHOUT global RoPE-table loads.
// tools/k64_trace_bench.py on the bench workload (profiles/r05_k64_trace_before.txt): of a wave's ~35 k cycles per 32-row tile the
// MFMAs take 5 k, the four 64-column slabs 30 k - each slab's 8 RoPE-table loads sit BEHIND the previous slab's 8 plane stores in the
// in-order VMEM queue (one vmcnt), so every slab waits for the previous slab's stores to retire.  Here the only VMEM traffic of a tile
// is the next tile's A fragments (issued right behind the MFMAs, i.e. OLDER than every store of the tile) and the plane stores:
//   * the axial RoPE table in its compact form (64 x-rows + 64 y-rows x 64 complex pairs = 64 KiB, GemmSplitArgs::rope_w) sits in LDS
//     next to the weights, rows XOR-swizzled by (row & 3) so that the four rows a lane quad reads do not collide;
//   * the accumulators go from "lane = column, registers = rows" to "lane i of a quad = row i, 4 consecutive columns" by a 4 x 4
//     transpose inside every lane quad (two DPP butterfly stages, as in gemm_x4g): the two complex pairs of a lane are its own, the
//     rotation needs no neighbour, and a store instruction writes 8 rows x 64 B (fp16) with nothing parked in LDS.
// Same arithmetic, same order per element: bit-identical to k_gemm_split_k64<8, true, *> (the memory attention at bench size
// was unchanged to the bit against the slab epilogue in round 5, profiles/HISTORY.md).
template <int CTL>
__device__ __forceinline__ float dpp_quad(float x) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTL, 0xf, 0xf, false));
}
__device__ __forceinline__ void quad_transpose4(float (&v)[4], bool b1, bool b2) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {          // lane ^ 2: register pairs (0, 2), (1, 3)
    const float a = v[j], b = v[j + 2];
    const float y = dpp_quad<0x4E>(b2 ? a : b);
    v[j] = b2 ? y : a;
    v[j + 2] = b2 ? b : y;
  }
#pragma unroll
  for (int j = 0; j < 4; j += 2) {       // lane ^ 1: register pairs (0, 1), (2, 3)
    const float a = v[j], b = v[j + 1];
    const float y = dpp_quad<0xB1>(b1 ? a : b);
    v[j] = b1 ? y : a;
    v[j + 1] = b1 ? b : y;
  }
}

constexpr int ROPE_LDS = 65536;   // [x | y part][64 rows][32 chunks of 16 B = 2 complex pairs]

template <bool F16H>
__global__ __launch_bounds__(512, 1) void k_gemm_split_k64t(GemmSplitArgs g, int tiles_per_wave) {
  constexpr int NT = 8, NCOL = 256, WPL = NCOL * WROWB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WPL + ROPE_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  for (int i = tid; i < NCOL * 8 * 2; i += 512) {   // weights -> LDS once (the image of k_gemm_split_k64)
    const int p = i / (NCOL * 8), r = (i / 8) % NCOL, c = i & 7;
    const uint4 v = *reinterpret_cast<const uint4*>((p ? g.W_lo : g.W_hi) + (size_t)r * g.ldw + c * 8);
    *reinterpret_cast<uint4*>(lds + p * WPL + r * WROWB + ((c ^ ((r >> 1) & 7)) << 4)) = v;
  }
  unsigned char* rope = lds + 2 * WPL;
  for (int i = tid; i < 2 * 64 * 32; i += 512) {    // compact axial table: pairs < 64 depend on x = pos % w, the others on y = pos / w
    const int part = i >> 11, r = (i >> 5) & 63, c = i & 31;
    const int pos = part ? r * g.rope_w : r;
    const float4 v = *reinterpret_cast<const float4*>(g.rope_cis + ((size_t)pos * 128 + part * 64 + 2 * c) * 2);
    *reinterpret_cast<float4*>(rope + part * 32768 + r * 512 + ((c ^ (r & 3)) << 4)) = v;
  }
  __syncthreads();

  const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
  const int ntiles = (g.M + 31) / 32;
  const int qi = l31 & 3, qq = l31 >> 2;
  const bool b1 = (lane & 1) != 0, b2 = (lane & 2) != 0;
  float bias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bias[t] = g.bias ? g.bias[t * 32 + l31] : 0.f;

  auto load_a = [&](int tile, AFrags& F) {
    int row = tile * 32 + l31;
    row = row < g.M ? row : g.M - 1;
    const unsigned short* ph = g.A_hi + (size_t)row * g.lda + half * 8;
    const unsigned short* pl = g.A_lo + (size_t)row * g.lda + half * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      F.h[s] = *reinterpret_cast<const bf16x8*>(ph + s * 16);
      F.l[s] = *reinterpret_cast<const bf16x8*>(pl + s * 16);
    }
  };
  AFrags F;
  load_a(gw < ntiles ? gw : ntiles - 1, F);
  const int wsw = (l31 >> 1) & 7;
  for (int it = 0; it < tiles_per_wave; ++it) {
    const int tile_raw = gw + it * nw;
    const int tile = tile_raw < ntiles ? tile_raw : ntiles - 1;
    const bool live = tile_raw < ntiles;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int coff = (((s * 2 + half) ^ wsw) << 4);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned char* wp = lds + (t * 32 + l31) * WROWB + coff;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wp);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wp + WPL);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.l[s], bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.h[s], bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.h[s], bh, acc[t], 0, 0, 0);
      }
    }
    {   // the next row tile's fragments: behind the MFMAs, ahead of every store of this tile
      const int tn_raw = gw + (it + 1) * nw;
      load_a(tn_raw < ntiles ? tn_raw : ntiles - 1, F);
    }
    const int m0 = tile * 32;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int m = m0 + 8 * gq + 4 * half + qi;          // this lane's output row of the group
      const bool row_ok = live && m < g.M;
      const int mm = m < g.M ? m : g.M - 1;
      const int tpos = mm % g.rope_L;
      const bool rot = tpos < g.rope_n;                   // (object-pointer tokens are not rotated)
      const int pos = rot ? tpos % g.rope_grid : 0;
      const int px = pos % g.rope_w, py = pos / g.rope_w;
      const unsigned char* rx = rope + px * 512;
      const unsigned char* ry = rope + 32768 + py * 512;
      unsigned short* const orow = g.C_hi + (size_t)mm * g.ldcp + 4 * qq;
      unsigned short* const lrow = F16H ? nullptr : g.C_lo + (size_t)mm * g.ldcp + 4 * qq;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = acc[t][4 * gq + k] + bias[t];
        quad_transpose4(v, b1, b2);
        {   // apply_rotary_enc (position_encoding.py:196-220) on the complex pairs (n, n+1), (n+2, n+3), n = 32 t + 4 q
#pragma clang fp contract(off)
          const int c = (t & 3) * 8 + qq;                 // 16-byte chunk of the part's row: pairs 16 (t & 3) + 2 q, + 1
          const unsigned char* rp = t < 4 ? rx + ((c ^ (px & 3)) << 4) : ry + ((c ^ (py & 3)) << 4);
          const float4 cs = *reinterpret_cast<const float4*>(rp);
          // (the contraction hipcc chose for `v0 c.x - v1 c.y`, `v0 c.y + v1 c.x` in the slab epilogue - read off its v_pk_mul / v_pk_fma
          //  pair: the products with c.y are rounded, the ones with c.x fused)
          const float a0 = __builtin_fmaf(v[0], cs.x, -(v[1] * cs.y)), a1 = __builtin_fmaf(v[1], cs.x, v[0] * cs.y);
          const float a2 = __builtin_fmaf(v[2], cs.z, -(v[3] * cs.w)), a3 = __builtin_fmaf(v[3], cs.z, v[2] * cs.w);
          v[0] = rot ? a0 : v[0]; v[1] = rot ? a1 : v[1]; v[2] = rot ? a2 : v[2]; v[3] = rot ? a3 : v[3];
        }
        if (row_ok) {
          if constexpr (F16H) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            uint2 h;
            h.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{ds2_sat_f16(v[0]), ds2_sat_f16(v[1])}), f16x2));
            h.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{ds2_sat_f16(v[2]), ds2_sat_f16(v[3])}), f16x2));
            *reinterpret_cast<uint2*>(orow + t * 32) = h;
          } else {
            uint2 h, l;
            h.x = cvt_pk_bf16(v[0], v[1]);
            h.y = cvt_pk_bf16(v[2], v[3]);
            l.x = cvt_pk_bf16(v[0] - bf_lo(h.x), v[1] - bf_hi(h.x));
            l.y = cvt_pk_bf16(v[2] - bf_lo(h.y), v[3] - bf_hi(h.y));
            *reinterpret_cast<uint2*>(orow + t * 32) = h;
            if (g.C_lo) *reinterpret_cast<uint2*>(lrow + t * 32) = l;
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // (unit by unit: left alone, hipcc hoists the 32 table reads of a tile and spills 200 registers)
      }
    }
  }
}

}  // namespace

#ifdef DS2_K64_TRACE
extern "C" int ds2_debug_k64_trace(unsigned long long* out) {   // [8][256] host buffer
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k64_trace), sizeof(unsigned long long) * 8 * 256) == hipSuccess ? 0 : 1;
}
#endif

bool gemm_split_k64_supported(const GemmSplitArgs& g) {
  const int ncols = g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N;
  return g.Kp == KW && (ncols == 128 || ncols == 256) && g.N > ncols - 32 && g.M >= 4096;
}

int launch_gemm_split_k64(const GemmSplitArgs& g, hipStream_t st) {
  DS2_REQUIRE(gemm_split_k64_supported(g), "gemm_split_k64: unsupported shape");
  const int ncols = g.C_hi ? (g.ldcp > g.N ? g.ldcp : g.N) : g.N;
  const int ntiles = (g.M + 31) / 32;
  int blocks = (ntiles + 7) / 8;
  if (blocks > 256) blocks = 256;                      // one persistent workgroup per CU
  const int tiles_per_wave = (ntiles + blocks * 8 - 1) / (blocks * 8);
  const bool planes = g.act == DS2_ACT_NONE && !g.gamma && !g.R && !g.C && g.C_hi;
  DS2_REQUIRE(!g.c_hi_f16 || (planes && !g.C_lo), "gemm_split_k64: fp16 hi plane needs the plane-only epilogue without a lo plane");
  // the key projection (256 columns, planes only, axial RoPE with the compact table): the register-transposed epilogue
  const bool k64t = planes && ncols == 256 && g.N == 256 && g.ldcp == 256 && g.rope_cis && g.rope_w == 64 &&
                    g.rope_grid == 4096 && g.rope_L > 0 && (g.c_hi_f16 || g.C_lo);
  if (k64t && g.c_hi_f16)
    hipLaunchKernelGGL((k_gemm_split_k64t<true>), dim3(blocks), dim3(512), 0, st, g, tiles_per_wave);
  else if (k64t)
    hipLaunchKernelGGL((k_gemm_split_k64t<false>), dim3(blocks), dim3(512), 0, st, g, tiles_per_wave);
  else if (g.c_hi_f16 && ncols == 256)
    hipLaunchKernelGGL((k_gemm_split_k64<8, true, true>), dim3(blocks), dim3(512), 0, st, g, tiles_per_wave);
  else if (g.c_hi_f16)
    hipLaunchKernelGGL((k_gemm_split_k64<4, true, true>), dim3(blocks), dim3(512), 0, st, g, tiles_per_wave);
  else if (ncols == 256 && planes)
    hipLaunchKernelGGL((k_gemm_split_k64<8, true>), dim3(blocks), dim3(512), 0, st, g, tiles_per_wave);
  else if (ncols == 256)
    hipLaunchKernelGGL((k_gemm_split_k64<8, false>), dim3(blocks), dim3(512), 0, st, g, tiles_per_wave);
  else if (planes)
    hipLaunchKernelGGL((k_gemm_split_k64<4, true>), dim3(blocks), dim3(512), 0, st, g, tiles_per_wave);
  else
    hipLaunchKernelGGL((k_gemm_split_k64<4, false>), dim3(blocks), dim3(512), 0, st, g, tiles_per_wave);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
