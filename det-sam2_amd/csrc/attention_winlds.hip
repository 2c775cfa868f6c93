// Hiera WINDOW attention with the whole window resident in LDS (round 4): 16 x 16 windows (256 tokens), head dim 72 - the 33
// windowed blocks of hiera_l's stage 3 (window_spec (8, 4, 16, 8); MultiScaleAttention inside window_partition,
// sam2/modeling/backbones/hieradet.py:46-93, 132-168; pad tokens of border windows carry the qkv bias as their key / value, as in
// the reference's zero-padded input) - and 14 x 14 windows at the same head dim (the other configs' window size; their head dims
// 96 / 112 do not fit: 2 x 86 KiB).
//
// Why: k_attention_bf16x3 fetches, splits and stages every 32-key tile inside the key loop with all eight waves in the same phase
// and a barrier per tile (MFMA busy ~ 16 %, SQ_WAIT_ANY 57 % of its wave cycles: profiles/r04_pmc_by_kernel_sq.txt).  A window's
// keys and values are used by all of its queries and fit in LDS once split: K planes 2 x 256 rows x 160 B = 80 KiB, V^T
// planes 2 x 72 rows x 512 B = 72 KiB.  One workgroup (8 waves) per (window, head): phase 1 stages K and V^T of the window once
// (16-byte LDS stores, conflict-free XOR-swizzled rows without pad bytes); phase 2 has NO barrier - each wave walks the key
// tiles for its 32 queries on its own, so the two waves of a SIMD drift apart and one's softmax hides behind the
// other's MFMAs.  Same arithmetic as k_attention_bf16x3 (three bf16 terms in the same order, online softmax over the same tiles
// in the same order).  Measured (hiera_l, 16-frame launches, 2048 workgroups): 290 -> 190 us; staging alone 78 us = the 453 MB of
// q / k / v read once at 5.8 TB/s, the key loops alone 117 us - the two phases of a CU do not overlap (one workgroup per CU), the
// next step would be a persistent workgroup that fetches the next window's rows into registers under the current key loop.
// Also measured, no gain: K fragments of the next tile / V^T fragments requested ahead of their MFMAs (195 us), a half-tile
// start offset between the two waves of a SIMD.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void split8(const float* v, bf16x8& p0, bf16x8& p1) {
  uint4 h, l;
  h.x = cvt_pk_bf16(v[0], v[1]); h.y = cvt_pk_bf16(v[2], v[3]);
  h.z = cvt_pk_bf16(v[4], v[5]); h.w = cvt_pk_bf16(v[6], v[7]);
  l.x = cvt_pk_bf16(v[0] - bf_lo(h.x), v[1] - bf_hi(h.x));
  l.y = cvt_pk_bf16(v[2] - bf_lo(h.y), v[3] - bf_hi(h.y));
  l.z = cvt_pk_bf16(v[4] - bf_lo(h.z), v[5] - bf_hi(h.z));
  l.w = cvt_pk_bf16(v[6] - bf_lo(h.w), v[7] - bf_hi(h.w));
  p0 = __builtin_bit_cast(bf16x8, h);
  p1 = __builtin_bit_cast(bf16x8, l);
}

constexpr int D = 72, DP = 80, KS = DP / 16, NT = (D + 31) / 32;   // head dim, padded to the MFMA depth; 3 dv blocks
constexpr int KROW = DP * 2;                                        // bytes per K row (one plane)
constexpr int OFF_K = 1024;

template <int WIN>
struct WinGeom {
  static constexpr int NTOK = WIN * WIN, NKT = (NTOK + 31) / 32, NKP = NKT * 32;   // tokens, key tiles, key slots (14: 196 / 7 / 224)
  static constexpr int VROW = NKP * 2;                                              // bytes per V^T row (one plane)
  static constexpr int KPLANE = NKP * KROW, VPLANE = D * VROW;
  static constexpr int OFF_V = OFF_K + 2 * KPLANE;
  // V^T rows 72..95 of the third dv block are never stored: their products land in output rows that are never written, the
  // lanes that would read them read row dv - 32 instead
  static constexpr int LDS_BYTES = OFF_V + 2 * VPLANE;
  static_assert(LDS_BYTES <= 160 * 1024, "window does not fit in LDS");
  // XOR of a V^T row's 16-byte chunk index: rows of 512 bytes all start at bank 0 (16 lanes need 16 different chunks), rows of
  // 448 bytes repeat every 4 rows
  __host__ __device__ static constexpr int vsw(int dv) { return WIN == 16 ? (dv & 15) : ((dv >> 2) & 3); }
};

template <int WIN>
__global__ __launch_bounds__(512) void k_attention_winlds(AttnArgs a) {
  using G = WinGeom<WIN>;
  constexpr int NTOK = G::NTOK, NKT = G::NKT, NKP = G::NKP, VROW = G::VROW, KPLANE = G::KPLANE, VPLANE = G::VPLANE, OFF_V = G::OFF_V;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int* rowtab = reinterpret_cast<int*>(lds);   // window-local token -> global row (-1: pad token of a border window, -2: no token)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.y, h = blockIdx.x;

  if (tid < NKP) {
    int row = -2;
    if (tid < NTOK) {
      int bw = b;
      long base = 0;
      if (a.wins > 0) { const int img = b / a.wins; bw = b - img * a.wins; base = (long)img * a.Hk * a.Wk; }
      const int wy = bw / a.nwx, wx = bw - wy * a.nwx;
      const int ly = tid / WIN, lx = tid - ly * WIN;
      const int y = wy * WIN + ly, x = wx * WIN + lx;
      row = (y < a.Hk && x < a.Wk) ? (int)(base + (long)y * a.Wk + x) : -1;
    }
    rowtab[tid] = row;
  }
  __syncthreads();

  // ---- phase 1a: K planes.  Item = (key slot r, 16-byte chunk c of its row): 8 columns split into hi / lo, chunk stored at
  // c ^ ((r >> 3) & 1) - with 160-byte rows the 16 lanes of a b128 read group then hit 16 different bank quads
  // (both staging loops have compile-time trip counts and are fully unrolled: all global loads of a thread are in flight
  // before the first split - a rolled loop pays one memory latency per iteration)
  constexpr int NKI = (NKP * (DP / 8) + 511) / 512;
  float kv[NKI][8];
#pragma unroll
  for (int i = 0; i < NKI; ++i) {
    const int it = tid + i * 512;
    const int r = it / (DP / 8), c = it - r * (DP / 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) kv[i][j] = 0.f;
    if (it < NKP * (DP / 8) && c < D / 8) {
      const int row = rowtab[r];
      const float* p = row >= 0 ? a.k + (size_t)row * a.ldk + h * D : ((row == -1 && a.k_pad) ? a.k_pad + h * D : nullptr);
      if (p) {
        const float4 x0 = *reinterpret_cast<const float4*>(p + c * 8), x1 = *reinterpret_cast<const float4*>(p + c * 8 + 4);
        kv[i][0] = x0.x; kv[i][1] = x0.y; kv[i][2] = x0.z; kv[i][3] = x0.w; kv[i][4] = x1.x; kv[i][5] = x1.y; kv[i][6] = x1.z; kv[i][7] = x1.w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NKI; ++i) {
    const int it = tid + i * 512;
    if (it < NKP * (DP / 8)) {
      const int r = it / (DP / 8), c = it - r * (DP / 8);
      bf16x8 hi, lo;
      split8(kv[i], hi, lo);
      const int off = r * KROW + ((c ^ ((r >> 3) & 1)) << 4);
      *reinterpret_cast<bf16x8*>(lds + OFF_K + off) = hi;
      *reinterpret_cast<bf16x8*>(lds + OFF_K + KPLANE + off) = lo;
    }
  }
  // ---- phase 1b: V^T planes.  Item = (dv, octet o of key positions): o = 4 kt + 2 s + hh holds the keys the lanes of half hh
  // feed to P.V step s of tile kt (position p = 16 s + 8 hh + j <-> accumulator row r = 8 s + j <-> key (r & 3) + 8 (r >> 2) + 4 hh);
  // lanes run along dv (coalesced global reads), chunk stored at o ^ ((dv >> 2) & 3)
  constexpr int NVI = (D * (NKP / 8) + 511) / 512;
  float vv[NVI][8];
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int it = tid + i * 512;
    const int dv = it % D, o = it / D;
    const int kt = o >> 2, s = (o >> 1) & 1, hh = o & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      vv[i][j] = 0.f;
      if (it < D * (NKP / 8)) {
        const int r = 8 * s + j;
        const int row = rowtab[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
        const float* p = row >= 0 ? a.v + (size_t)row * a.ldv + h * D : ((row == -1 && a.v_pad) ? a.v_pad + h * D : nullptr);
        if (p) vv[i][j] = p[dv];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int it = tid + i * 512;
    if (it < D * (NKP / 8)) {
      const int dv = it % D, o = it / D;
      bf16x8 hi, lo;
      split8(vv[i], hi, lo);
      const int off = dv * VROW + ((o ^ G::vsw(dv)) << 4);
      *reinterpret_cast<bf16x8*>(lds + OFF_V + off) = hi;
      *reinterpret_cast<bf16x8*>(lds + OFF_V + VPLANE + off) = lo;
    }
  }

  // ---- Q of this wave's 32 queries: straight from global into the B-operand fragments (scaled, split)
  const int qi = wave * 32 + l31;
  const bool wave_active = wave * 32 < NTOK;
  const int qrow = (wave_active && qi < NTOK) ? rowtab[qi] : -2;
  bf16x8 q0[KS], q1[KS];
  {
    const float sc = a.scale * 1.44269504088896340736f;
    const float* qp = qrow >= 0 ? a.q + (size_t)qrow * a.ldq + h * D : nullptr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int c0 = ks * 16 + half * 8;
      if (qp && c0 < D) {
        const float4 x0 = *reinterpret_cast<const float4*>(qp + c0), x1 = *reinterpret_cast<const float4*>(qp + c0 + 4);
        v[0] = x0.x * sc; v[1] = x0.y * sc; v[2] = x0.z * sc; v[3] = x0.w * sc;
        v[4] = x1.x * sc; v[5] = x1.y * sc; v[6] = x1.z * sc; v[7] = x1.w * sc;
      }
      split8(v, q0[ks], q1[ks]);
    }
  }
  __syncthreads();
  if (!wave_active) return;
  // (measured and not kept: starting waves 4-7 half a tile late so that the two waves of a SIMD run out of phase - 0 ... 32 x 64
  // clocks of s_sleep all give the same time)

  // ---- phase 2: no barrier from here on
  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const unsigned char* kp0 = lds + OFF_K + l31 * KROW;
  const int ksw = (l31 >> 3) & 1;                     // (rows kt * 32 + l31: bit 3 of the row is bit 3 of l31)
#pragma unroll 1
  for (int kt = 0; kt < NKT; ++kt) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const unsigned char* kb = kp0 + kt * 32 * KROW;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int co = ((2 * ks + half) ^ ksw) << 4;
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kb + co);
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kb + KPLANE + co);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q0[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q1[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0[ks], acc, 0, 0, 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + mfma32_row(e, half);
      if (NTOK < NKP) acc[e] = key < NTOK ? acc[e] : -INFINITY;    // scores already carry scale * log2(e) (folded into Q)
      tmax = fmaxf(tmax, acc[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
    float p[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      p[e] = __builtin_amdgcn_exp2f(acc[e] - m_new);
      psum += p[e];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    bf16x8 pb0[2], pb1[2];
    split8(p, pb0[0], pb1[0]);
    split8(p + 8, pb0[1], pb1[1]);
    if (__any(alpha != 1.f)) {   // (exact: alpha == 1 leaves o unchanged)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int dvr = t * 32 + l31 < D ? t * 32 + l31 : t * 32 + l31 - 32;
      const unsigned char* vb = lds + OFF_V + dvr * VROW;
      const int vsw = G::vsw(dvr);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int co = ((kt * 4 + s * 2 + half) ^ vsw) << 4;
        const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vb + co);
        const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(vb + VPLANE + co);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb0[s], o[t], 0, 0, 0);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb1[s], o[t], 0, 0, 0);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb0[s], o[t], 0, 0, 0);
      }
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  if (qrow < 0) return;   // pad tokens of a border window and the slots beyond the 196th token have no output row
  const size_t orow = (size_t)qrow;
  if (a.o_hi) {   // registers 4k..4k+3 of a fragment are 4 consecutive output columns
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int dv = t * 32 + 8 * k + 4 * half;
        if (dv < D) {
#pragma clang fp contract(off)
          const float v0 = o[t][4 * k] * inv, v1 = o[t][4 * k + 1] * inv, v2 = o[t][4 * k + 2] * inv, v3 = o[t][4 * k + 3] * inv;
          uint2 hh, ll;
          if (a.o_mx) {   // "MX" activation planes (common.h)
            ds2_mx_pair(v0, v1, false, hh.x, ll.x);
            ds2_mx_pair(v2, v3, false, hh.y, ll.y);
          } else {
            hh.x = cvt_pk_bf16(v0, v1);
            hh.y = cvt_pk_bf16(v2, v3);
            ll.x = cvt_pk_bf16(v0 - bf_lo(hh.x), v1 - bf_hi(hh.x));
            ll.y = cvt_pk_bf16(v2 - bf_lo(hh.y), v3 - bf_hi(hh.y));
          }
          *reinterpret_cast<uint2*>(a.o_hi + orow * a.ldop + h * D + dv) = hh;
          *reinterpret_cast<uint2*>(a.o_lo + orow * a.ldop + h * D + dv) = ll;
        }
      }
  } else {
    float* op = a.o + orow * a.ldo + h * D;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int dv = t * 32 + 8 * k + 4 * half;
        if (dv < D)
          *reinterpret_cast<float4*>(op + dv) = make_float4(o[t][4 * k] * inv, o[t][4 * k + 1] * inv, o[t][4 * k + 2] * inv, o[t][4 * k + 3] * inv);
      }
  }
}


// ---- two-phase form for head dims whose K and V^T planes do not fit TOGETHER (hiera_t / hiera_s: 96; 14 x 14 windows):
// K planes resident -> all scores of a wave's 32 queries in registers (7 tiles x 16) -> exact softmax over the whole window (no
// running maximum, no rescale) -> the SAME LDS region re-filled with the V^T planes -> P.V.  Three barriers per workgroup, none
// inside a phase; the V rows are fetched into registers behind the score phase so that their latency falls under the softmax.
template <int DH>
struct Win2pGeom {
  static constexpr int WIN = 14, NTOK = WIN * WIN, NKT = (NTOK + 31) / 32, NKP = NKT * 32;
  static constexpr int KS = DH / 16, NT = (DH + 31) / 32;
  static constexpr int KROW2 = DH * 2, VROW = NKP * 2;
  static constexpr int KPLANE = NKP * KROW2, VPLANE = NT * 32 * VROW;
  static constexpr int REGION = 2 * (KPLANE > VPLANE ? KPLANE : VPLANE);
  static constexpr int LDS_BYTES = OFF_K + REGION;
  static_assert(DH % 16 == 0 && LDS_BYTES <= 160 * 1024, "head dim not supported by the two-phase window kernel");
  // K rows of 192 bytes (48 banks) repeat their bank every 4 rows, rows of 224 bytes (56 banks) every 8
  __host__ __device__ static constexpr int ksw(int r) { return DH == 96 ? ((r >> 2) & 3) : ((r >> 3) & 1); }
  __host__ __device__ static constexpr int vsw(int dv) { return (dv >> 2) & 3; }
};

template <int DH>
__global__ __launch_bounds__(512) void k_attention_win2p(AttnArgs a) {
  using G = Win2pGeom<DH>;
  constexpr int WIN = G::WIN, NTOK = G::NTOK, NKT = G::NKT, NKP = G::NKP, KS = G::KS, NT = G::NT;
  constexpr int KROW2 = G::KROW2, VROW = G::VROW, KPLANE = G::KPLANE, VPLANE = G::VPLANE;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int* rowtab = reinterpret_cast<int*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.y, h = blockIdx.x;

  if (tid < NKP) {
    int row = -2;
    if (tid < NTOK) {
      int bw = b;
      long base = 0;
      if (a.wins > 0) { const int img = b / a.wins; bw = b - img * a.wins; base = (long)img * a.Hk * a.Wk; }
      const int wy = bw / a.nwx, wx = bw - wy * a.nwx;
      const int ly = tid / WIN, lx = tid - ly * WIN;
      const int y = wy * WIN + ly, x = wx * WIN + lx;
      row = (y < a.Hk && x < a.Wk) ? (int)(base + (long)y * a.Wk + x) : -1;
    }
    rowtab[tid] = row;
  }
  __syncthreads();

  // ---- K planes (as k_attention_winlds)
  {
    constexpr int CH = DH / 8, NKI = (NKP * CH + 511) / 512;
    float kv[NKI][8];
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const int it = tid + i * 512;
      const int r = it / CH, c = it - r * CH;
#pragma unroll
      for (int j = 0; j < 8; ++j) kv[i][j] = 0.f;
      if (it < NKP * CH) {
        const int row = rowtab[r];
        const float* p = row >= 0 ? a.k + (size_t)row * a.ldk + h * DH : ((row == -1 && a.k_pad) ? a.k_pad + h * DH : nullptr);
        if (p) {
          const float4 x0 = *reinterpret_cast<const float4*>(p + c * 8), x1 = *reinterpret_cast<const float4*>(p + c * 8 + 4);
          kv[i][0] = x0.x; kv[i][1] = x0.y; kv[i][2] = x0.z; kv[i][3] = x0.w; kv[i][4] = x1.x; kv[i][5] = x1.y; kv[i][6] = x1.z; kv[i][7] = x1.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const int it = tid + i * 512;
      if (it < NKP * CH) {
        const int r = it / CH, c = it - r * CH;
        bf16x8 hi, lo;
        split8(kv[i], hi, lo);
        const int off = r * KROW2 + ((c ^ G::ksw(r)) << 4);
        *reinterpret_cast<bf16x8*>(lds + OFF_K + off) = hi;
        *reinterpret_cast<bf16x8*>(lds + OFF_K + KPLANE + off) = lo;
      }
    }
  }
  // ---- Q of this wave's 32 queries
  const int qi = wave * 32 + l31;
  const bool wave_active = wave * 32 < NTOK;
  const int qrow = (wave_active && qi < NTOK) ? rowtab[qi] : -2;
  f32x16 S[NKT];
  {
    bf16x8 q0[KS], q1[KS];
    const float sc = a.scale * 1.44269504088896340736f;
    const float* qp = qrow >= 0 ? a.q + (size_t)qrow * a.ldq + h * DH : nullptr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (qp) {
        const float4 x0 = *reinterpret_cast<const float4*>(qp + ks * 16 + half * 8), x1 = *reinterpret_cast<const float4*>(qp + ks * 16 + half * 8 + 4);
        v[0] = x0.x * sc; v[1] = x0.y * sc; v[2] = x0.z * sc; v[3] = x0.w * sc;
        v[4] = x1.x * sc; v[5] = x1.y * sc; v[6] = x1.z * sc; v[7] = x1.w * sc;
      }
      split8(v, q0[ks], q1[ks]);
    }
    __syncthreads();
    // ---- phase 1: all scores of the window (no barrier inside)
    if (wave_active) {
      const int ksw = G::ksw(l31);          // (row kt * 32 + l31: the swizzle bits are bits of l31)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const unsigned char* kb = lds + OFF_K + (kt * 32 + l31) * KROW2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int co = ((2 * ks + half) ^ ksw) << 4;
          const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kb + co);
          const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kb + KPLANE + co);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q0[ks], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q1[ks], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0[ks], acc, 0, 0, 0);
        }
        S[kt] = acc;
      }
    }
  }
  // ---- V rows of the window into registers (all threads), consumed after the softmax
  constexpr int NVI = (DH * (NKP / 8) + 511) / 512;
  float vv[NVI][8];
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int it = tid + i * 512;
    const int dv = it % DH, o = it / DH;
    const int kt = o >> 2, s = (o >> 1) & 1, hh = o & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      vv[i][j] = 0.f;
      if (it < DH * (NKP / 8)) {
        const int r = 8 * s + j;
        const int row = rowtab[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
        const float* p = row >= 0 ? a.v + (size_t)row * a.ldv + h * DH : ((row == -1 && a.v_pad) ? a.v_pad + h * DH : nullptr);
        if (p) vv[i][j] = p[dv];
      }
    }
  }
  // ---- exact softmax over the window's keys: maximum, exponentials (in place), sum
  float l_tot = 1.f;
  if (wave_active) {
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (kt * 32 + mfma32_row(e, half) >= NTOK) S[kt][e] = -INFINITY;   // (compile-time: only the last tile has such rows)
        m = fmaxf(m, S[kt][e]);
      }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        S[kt][e] = __builtin_amdgcn_exp2f(S[kt][e] - m);
        l += S[kt][e];
      }
    l_tot = l + __shfl_xor(l, 32);
  }
  __syncthreads();   // every wave is done with the K planes
  // ---- V^T planes into the same region
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int it = tid + i * 512;
    if (it < DH * (NKP / 8)) {
      const int dv = it % DH, o = it / DH;
      bf16x8 hi, lo;
      split8(vv[i], hi, lo);
      const int off = dv * VROW + ((o ^ G::vsw(dv)) << 4);
      *reinterpret_cast<bf16x8*>(lds + OFF_K + off) = hi;
      *reinterpret_cast<bf16x8*>(lds + OFF_K + VPLANE + off) = lo;
    }
  }
  __syncthreads();
  if (!wave_active) return;   // (whole waves only: the MFMAs below want every lane of the wave)

  // ---- phase 2: O^T = V^T P^T
  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    bf16x8 pb0[2], pb1[2];
    float p[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) p[e] = S[kt][e];
    split8(p, pb0[0], pb1[0]);
    split8(p + 8, pb0[1], pb1[1]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int dvr = t * 32 + l31 < DH ? t * 32 + l31 : t * 32 + l31 - 32;
      const unsigned char* vb = lds + OFF_K + dvr * VROW;
      const int vsw = G::vsw(dvr);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int co = ((kt * 4 + s * 2 + half) ^ vsw) << 4;
        const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vb + co);
        const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(vb + VPLANE + co);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb0[s], o[t], 0, 0, 0);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb1[s], o[t], 0, 0, 0);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb0[s], o[t], 0, 0, 0);
      }
    }
  }
  const float inv = 1.f / l_tot;
  if (qrow < 0) return;   // pad tokens of a border window and the slots beyond the 196th token have no output row
  const size_t orow = (size_t)qrow;
  if (a.o_hi) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int dv = t * 32 + 8 * k + 4 * half;
        if (dv < DH) {
#pragma clang fp contract(off)
          const float v0 = o[t][4 * k] * inv, v1 = o[t][4 * k + 1] * inv, v2 = o[t][4 * k + 2] * inv, v3 = o[t][4 * k + 3] * inv;
          uint2 hh, ll;
          if (a.o_mx) {   // "MX" activation planes (common.h)
            ds2_mx_pair(v0, v1, false, hh.x, ll.x);
            ds2_mx_pair(v2, v3, false, hh.y, ll.y);
          } else {
            hh.x = cvt_pk_bf16(v0, v1);
            hh.y = cvt_pk_bf16(v2, v3);
            ll.x = cvt_pk_bf16(v0 - bf_lo(hh.x), v1 - bf_hi(hh.x));
            ll.y = cvt_pk_bf16(v2 - bf_lo(hh.y), v3 - bf_hi(hh.y));
          }
          *reinterpret_cast<uint2*>(a.o_hi + orow * a.ldop + h * DH + dv) = hh;
          *reinterpret_cast<uint2*>(a.o_lo + orow * a.ldop + h * DH + dv) = ll;
        }
      }
  } else {
    float* op = a.o + orow * a.ldo + h * DH;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int dv = t * 32 + 8 * k + 4 * half;
        if (dv < DH)
          *reinterpret_cast<float4*>(op + dv) = make_float4(o[t][4 * k] * inv, o[t][4 * k + 1] * inv, o[t][4 * k + 2] * inv, o[t][4 * k + 3] * inv);
      }
  }
}

}  // namespace

// 16 x 16 (hiera_l stage 3) or 14 x 14 windows of one or several images, queries and keys on the same grid, head dim 72
bool attention_winlds_supported(const AttnArgs& a) {
  const bool one_phase = a.D == D && (a.win_q == 16 || a.win_q == 14), two_phase = a.D == 96 && a.win_q == 14;
  return a.DV == a.D && (one_phase || two_phase) && a.win_k == a.win_q && a.Lq == a.win_q * a.win_q && a.Lk == a.Lq &&
         a.Hq == a.Hk && a.Wq == a.Wk &&
         a.nwx > 0 && a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && (a.o_hi ? a.ldop % 4 == 0 : a.ldo % 4 == 0) && a.heads <= 65535 &&
         a.batch <= 65535;
}

int launch_attention_winlds(const AttnArgs& a, hipStream_t st) {
  DS2_REQUIRE(attention_winlds_supported(a), "attention_winlds: unsupported arguments");
  static bool attr_done = false;
  if (!attr_done) {
    DS2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_winlds<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DS2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_winlds<14>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DS2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_win2p<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  if (a.D == 96) hipLaunchKernelGGL(k_attention_win2p<96>, dim3(a.heads, a.batch), dim3(512), Win2pGeom<96>::LDS_BYTES, st, a);
  else if (a.win_q == 16) hipLaunchKernelGGL(k_attention_winlds<16>, dim3(a.heads, a.batch), dim3(512), WinGeom<16>::LDS_BYTES, st, a);
  else hipLaunchKernelGGL(k_attention_winlds<14>, dim3(a.heads, a.batch), dim3(512), WinGeom<14>::LDS_BYTES, st, a);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
