// Split-precision ("bf16x3") flash attention on the CDNA4 bf16 matrix cores, same structure as
// attention.hip (transposed scores S^T = K Q^T so a lane owns one query column; P^T feeds the second
// product straight from the accumulator registers), with every fp32 operand split into two bf16 planes
//     x = x0 + x1,   product = a0*b0 + a0*b1 + a1*b0   (fp32 accumulate, ~2^-16 relative error),
// i.e. three v_mfma_f32_32x32x16_bf16 where attention.hip issues eight v_mfma_f32_32x32x2_f32 (5.3x less
// matrix-pipe time).  Splits happen once per element: Q when it is loaded into registers (pre-multiplied by
// scale*log2e), K and V when their 32-key tile is staged into LDS, P in registers after the softmax.
//
// LDS images (bf16):  K planes [32 keys][D] with a 16-byte row pad -> every lane's operand is ONE
// ds_read_b128 and the 16-lane read groups are conflict-free; V is stored TRANSPOSED [DV][32 keys] with the
// keys permuted into the accumulator's own row order (mfma32_row), so the V^T operand of O^T = V^T P^T is
// also one ds_read_b128 and P needs no cross-lane movement.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BKEYS = 32, VROWB = 80;   // V^T row: 32 keys * 2 B + 16 B pad

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// split 8 floats into two packed bf16x8 planes
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

#ifndef DS2_HATT_VEARLY
#define DS2_HATT_VEARLY 1
#endif
#ifndef DS2_HATT_V8
#define DS2_HATT_V8 0
#endif
// DS2_ATT_TRACE (profiling builds only): wave 0 / 4 of workgroup (0,0,0) stamp s_memtime along their key tiles; =1 traces the
// global-attention launches (Lk = 4096), =2 the 256-key window launches (tools/att_trace.py)
#ifdef DS2_ATT_TRACE
__device__ unsigned long long g_att_trace[2][512];
#define ATT_T()                                                                                     \
  if (trace_on && tix < 512) {                                                                      \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                     \
    if (lane == 0) g_att_trace[wave >> 2][tix] = t_;                                                \
    ++tix;                                                                                          \
  }
#else
#define ATT_T()
#endif

struct RowMap {
  int win, H, W, nwx, L, wins;
  __device__ __forceinline__ long row(int b, int i) const {
    if (win == 0) return (long)b * L + i;
    long base = 0;
    if (wins > 0) { const int img = b / wins; b -= img * wins; base = (long)img * H * W; }
    const int wy = b / nwx, wx = b - wy * nwx;
    const int ly = i / win, lx = i - ly * win;
    const int y = wy * win + ly, x = wx * win + lx;
    return (y < H && x < W) ? base + (long)y * W + x : -1;
  }
};

// position of key (0..31) inside a V^T row so that the 8 k-slots of MFMA k-step s read by half-wave h
// (keys mfma32_row(8s+j, h), j = 0..7) are contiguous: pos = 16s + 8h + j.
__device__ __forceinline__ int vt_pos(int key) {
  const int h = (key >> 2) & 1, r = (key & 3) + 4 * (key >> 3);
  return 16 * (r >> 3) + 8 * h + (r & 7);
}

// D / DV are the true head dims (multiples of 4); the MFMA shapes need DP = D rounded up to 16 and DVP = DV
// rounded up to 32: the pad columns/rows of the LDS images are zeroed once and never written again.
// NW = waves per workgroup (32 queries each): 8 where the query count allows it - a K / V tile is fetched, split and staged
// once per workgroup, so twice the queries per workgroup halve that work per query.
template <int D, int DV, int NW>
__global__ __launch_bounds__(64 * NW) void k_attention_bf16x3(AttnArgs a) {
  constexpr int NTHR = 64 * NW, BQ = 32 * NW;
  static_assert(D % 4 == 0 && DV % 4 == 0, "head dims must be multiples of 4");
  constexpr int DP = (D + 15) / 16 * 16, DVP = (DV + 31) / 32 * 32;
  constexpr int KS = DP / 16, NT = DVP / 32;
  constexpr int KROWB = DP * 2 + 16;                   // bytes per K row per plane
  constexpr int KPLANE = BKEYS * KROWB, VPLANE = DVP * VROWB;
  constexpr int NK4 = (8 * D + NTHR - 1) / NTHR, NV4 = (8 * DV + NTHR - 1) / NTHR;
  constexpr int QSLD = DP + 1;                          // fp32 Q staging row stride
  constexpr int QBYTES = 32 * QSLD * 4;
  constexpr int KBUF = 2 * KPLANE > QBYTES ? 2 * KPLANE : QBYTES;   // one K buffer doubles as the Q staging area
  __shared__ __attribute__((aligned(16))) unsigned char Kraw[2][KBUF];
  __shared__ __attribute__((aligned(16))) unsigned char Vp[2][2][VPLANE];
#define Kp(buf, plane, off) Kraw[buf][(plane) * KPLANE + (off)]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0i = blockIdx.x * BQ;
  const RowMap qm{a.win_q, a.Hq, a.Wq, a.nwx, a.Lq, a.wins};
  const RowMap km{a.win_k, a.Hk, a.Wk, a.nwx, a.Lk, a.wins};
  const float sc = a.scale * 1.44269504088896340736f;

  // ---- Q: stage fp32 rows through LDS, scale, split into two bf16 planes held in registers (B operand)
  bf16x8 q0[KS], q1[KS];
  {
    float* Qs = reinterpret_cast<float*>(&Kraw[0][0]);   // [32][DP+1] floats
    for (int idx = tid; idx < 32 * QSLD; idx += NTHR) Qs[idx] = 0.f;   // pad columns D..DP stay zero
    __syncthreads();
    for (int w = 0; w < NW; ++w) {
      for (int idx = tid; idx < 32 * (D / 4); idx += NTHR) {
        const int r = idx / (D / 4), c4 = idx - r * (D / 4);
        const int qi = q0i + w * 32 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qi < a.Lq) {
          const long row = qm.row(b, qi);
          if (row >= 0) v = *reinterpret_cast<const float4*>(a.q + row * a.ldq + h * D + c4 * 4);
        }
        float* dst = Qs + r * QSLD + c4 * 4;
        dst[0] = v.x * sc; dst[1] = v.y * sc; dst[2] = v.z * sc; dst[3] = v.w * sc;
      }
      __syncthreads();
      if (wave == w) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = Qs[l31 * QSLD + ks * 16 + half * 8 + j];
          split8(v, q0[ks], q1[ks]);
        }
      }
      __syncthreads();
    }
  }
  const bool wave_active = (q0i + wave * 32) < a.Lq;

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  float4 rk[NK4], rv[NV4];
  // DS2_HATT_V8: V is fetched per (dv, octet of V^T positions) - 8 dword loads along the key axis, lanes along dv (coalesced) -
  // and staged with ONE 16-byte LDS store per plane instead of eight 2-byte ones per float4 (tried for the "V stage" share of a tile, 11 % in the
  // trace of tools/att_trace.py; SLOWER - image encoder 9.21 -> 9.51 ms/frame: eight key cursors and eight 64-bit addresses per
  // slot cost more VALU than the 2-byte stores - so off by default)
  constexpr int NV8 = DS2_HATT_V8 ? (4 * DV + NTHR - 1) / NTHR : 0;
  float rv8[NV8 > 0 ? NV8 : 1][8];
  // Key rows.  The tiles are fetched in order, so each staging slot keeps a cursor (key index + position inside the key
  // window) that advances by one tile per fetch: no divisions in the loop (RowMap::row costs four per key; the trace of
  // tools/att_trace.py showed the address arithmetic of the fetches at 30-50 % of a windowed tile).
  struct KeyCur { int ki, ly, lx; };
  long kw_base = 0;
  int kw_y0 = 0, kw_x0 = 0;
  if (km.win > 0) {
    int bw = b;
    if (km.wins > 0) { const int img = b / km.wins; bw = b - img * km.wins; kw_base = (long)img * km.H * km.W; }
    const int wy = bw / km.nwx, wx = bw - wy * km.nwx;
    kw_y0 = wy * km.win; kw_x0 = wx * km.win;
  }
  const int kw_dy = km.win > 0 ? BKEYS / km.win : 0, kw_dx = km.win > 0 ? BKEYS % km.win : 0;
  auto cur_init = [&](int r) {
    KeyCur c{r, 0, 0};
    if (km.win > 0) { c.ly = r / km.win; c.lx = r - c.ly * km.win; }
    return c;
  };
  auto cur_row = [&](const KeyCur& c) -> long {   // = km.row(b, c.ki)
    if (km.win == 0) return (long)b * km.L + c.ki;
    const int y = kw_y0 + c.ly, x = kw_x0 + c.lx;
    return (y < km.H && x < km.W) ? kw_base + (long)y * km.W + x : -1;
  };
  auto cur_next = [&](KeyCur& c) {
    c.ki += BKEYS;
    c.ly += kw_dy; c.lx += kw_dx;
    if (km.win > 0 && c.lx >= km.win) { c.lx -= km.win; ++c.ly; }
  };
  KeyCur kcur[NK4], vcur[NV4];
#pragma unroll
  for (int i = 0; i < NK4; ++i) kcur[i] = cur_init((tid + NTHR * i) / (D / 4));
#pragma unroll
  for (int i = 0; i < NV4; ++i) vcur[i] = cur_init((tid + NTHR * i) / (DV / 4));
  auto load_k = [&](int) {   // fetches the NEXT tile of the sequence 0, 1, 2, ...
#pragma unroll
    for (int i = 0; i < NK4; ++i) {
      const int idx = tid + NTHR * i;
      rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < 8 * D) {
        const int c4 = idx % (D / 4);
        if (kcur[i].ki < a.Lk) {
          const long row = cur_row(kcur[i]);
          const float* p = row >= 0 ? a.k + row * a.ldk + h * D : (a.k_pad ? a.k_pad + h * D : nullptr);
          if (p) rk[i] = *reinterpret_cast<const float4*>(p + c4 * 4);
        }
        cur_next(kcur[i]);
      }
    }
  };
  KeyCur v8cur[NV8 > 0 ? NV8 : 1][8];
  if constexpr (DS2_HATT_V8) {
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
      const int oct = (tid + NTHR * i) / DV;            // position octet 0..3 = (s, h): keys 16 s + 4 h + {0..3, 8..11}
#pragma unroll
      for (int j = 0; j < 8; ++j) v8cur[i][j] = cur_init(16 * (oct >> 1) + 4 * (oct & 1) + (j & 3) + 8 * (j >> 2));
    }
  }
  auto load_v = [&](int) {
    if constexpr (DS2_HATT_V8) {
#pragma unroll
      for (int i = 0; i < NV8; ++i) {
        const int idx = tid + NTHR * i;
        const int dv = idx % DV;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rv8[i][j] = 0.f;
          if (idx < 4 * DV) {
            if (v8cur[i][j].ki < a.Lk) {
              const long row = cur_row(v8cur[i][j]);
              const float* p = row >= 0 ? a.v + row * a.ldv + h * DV : (a.v_pad ? a.v_pad + h * DV : nullptr);
              if (p) rv8[i][j] = p[dv];
            }
            cur_next(v8cur[i][j]);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int idx = tid + NTHR * i;
      rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < 8 * DV) {
        const int c4 = idx % (DV / 4);
        if (vcur[i].ki < a.Lk) {
          const long row = cur_row(vcur[i]);
          const float* p = row >= 0 ? a.v + row * a.ldv + h * DV : (a.v_pad ? a.v_pad + h * DV : nullptr);
          if (p) rv[i] = *reinterpret_cast<const float4*>(p + c4 * 4);
        }
        cur_next(vcur[i]);
      }
    }
  };
  auto store_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NK4; ++i) {
      const int idx = tid + NTHR * i;
      if (idx < 8 * D) {
        const int r = idx / (D / 4), c4 = idx - r * (D / 4);
        uint2 hi, lo;
        hi.x = cvt_pk_bf16(rk[i].x, rk[i].y);
        hi.y = cvt_pk_bf16(rk[i].z, rk[i].w);
        lo.x = cvt_pk_bf16(rk[i].x - bf_lo(hi.x), rk[i].y - bf_hi(hi.x));
        lo.y = cvt_pk_bf16(rk[i].z - bf_lo(hi.y), rk[i].w - bf_hi(hi.y));
        const int off = r * KROWB + c4 * 8;
        *reinterpret_cast<uint2*>(&Kp(buf, 0, off)) = hi;
        *reinterpret_cast<uint2*>(&Kp(buf, 1, off)) = lo;
      }
    }
  };
  auto store_v = [&](int buf) {   // transposed + key-permuted
    if constexpr (DS2_HATT_V8) {
#pragma unroll
      for (int i = 0; i < NV8; ++i) {
        const int idx = tid + NTHR * i;
        if (idx < 4 * DV) {
          const int dv = idx % DV, oct = idx / DV;
          bf16x8 hi, lo;
          split8(rv8[i], hi, lo);
          *reinterpret_cast<bf16x8*>(&Vp[buf][0][dv * VROWB + oct * 16]) = hi;
          *reinterpret_cast<bf16x8*>(&Vp[buf][1][dv * VROWB + oct * 16]) = lo;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int idx = tid + NTHR * i;
      if (idx < 8 * DV) {
        const int r = idx / (DV / 4), c4 = idx - r * (DV / 4);
        const int pos = vt_pos(r) * 2;
        const float v[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned hi = cvt_pk_bf16(v[j], 0.f);
          const unsigned lo = cvt_pk_bf16(v[j] - bf_lo(hi), 0.f);
          const int off = (c4 * 4 + j) * VROWB + pos;
          *reinterpret_cast<unsigned short*>(&Vp[buf][0][off]) = (unsigned short)(hi & 0xffffu);
          *reinterpret_cast<unsigned short*>(&Vp[buf][1][off]) = (unsigned short)(lo & 0xffffu);
        }
      }
    }
  };

  // zero the LDS images once: pad columns (D..DP of K rows) and pad rows (DV..DVP of V^T) are never written
  for (int idx = tid; idx < 2 * KBUF / 4; idx += NTHR) reinterpret_cast<unsigned*>(&Kraw[0][0])[idx] = 0u;
  for (int idx = tid; idx < 4 * VPLANE / 4; idx += NTHR) reinterpret_cast<unsigned*>(&Vp[0][0][0])[idx] = 0u;
  __syncthreads();
  const int nkt = (a.Lk + BKEYS - 1) / BKEYS;
  load_k(0);
  store_k(0);
  load_v(0);
  store_v(0);
  __syncthreads();
  int cur = 0;
#ifdef DS2_ATT_TRACE
  int tix = 0;
  const bool trace_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave & 3) == 0 && D == 72 &&
                        a.Lk == (DS2_ATT_TRACE == 1 ? 4096 : 256);
#endif
  for (int kt = 0; kt < nkt; ++kt) {
    ATT_T()   // 0: tile start
    if (kt + 1 < nkt) {
      load_k(kt + 1);
      if (DS2_HATT_VEARLY) load_v(kt + 1);   // V fetched with K: it is consumed 2 phases later, not right after its issue
    }
    ATT_T()   // 1: K loads issued
    f32x16 acc;
    bf16x8 pb0[2], pb1[2];
    float alpha = 1.f;
    if (wave_active) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      const unsigned char* k0p = &Kp(cur, 0, l31 * KROWB + half * 16);
      const unsigned char* k1p = &Kp(cur, 1, l31 * KROWB + half * 16);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(k0p + ks * 32);
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(k1p + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q0[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q1[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0[ks], acc, 0, 0, 0);
      }
      ATT_T()   // 2: QK MFMAs issued
      float tmax = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kt * BKEYS + mfma32_row(e, half);
        acc[e] = key < a.Lk ? acc[e] : -INFINITY;    // scores already carry scale*log2e (folded into Q)
        tmax = fmaxf(tmax, acc[e]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run, tmax);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.f;
      float p[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        p[e] = __builtin_amdgcn_exp2f(acc[e] - m_new);
        psum += p[e];
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
      split8(p, pb0[0], pb1[0]);
      split8(p + 8, pb0[1], pb1[1]);
    }
    ATT_T()   // 3: softmax + P split done
    if (kt + 1 < nkt) {
      store_k(cur ^ 1);
      if (!DS2_HATT_VEARLY) load_v(kt + 1);
    }
    ATT_T()   // 4: K staged, V loads issued
    if (wave_active) {
      if (__any(alpha != 1.f)) {   // (exact: alpha == 1 leaves o unchanged; after the first tiles the common case)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned char* v0p = &Vp[cur][0][(t * 32 + l31) * VROWB + half * 16];
        const unsigned char* v1p = &Vp[cur][1][(t * 32 + l31) * VROWB + half * 16];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(v0p + s * 32);
          const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(v1p + s * 32);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb0[s], o[t], 0, 0, 0);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb1[s], o[t], 0, 0, 0);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb0[s], o[t], 0, 0, 0);
        }
      }
    }
    ATT_T()   // 5: PV MFMAs issued
    if (kt + 1 < nkt) store_v(cur ^ 1);
    ATT_T()   // 6: V staged
    __syncthreads();
    cur ^= 1;
  }

  if (!wave_active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  const int qi = q0i + wave * 32 + l31;
  if (qi >= a.Lq) return;
  const long orow = qm.row(b, qi);
  if (orow < 0) return;
  if (a.o_hi) {   // registers 4k..4k+3 of a fragment are 4 consecutive output columns
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int dv = t * 32 + 8 * k + 4 * half;
        if (dv < DV) {
          // (no FMA contraction of `o * inv - hi`: the lo plane must be the remainder of the ROUNDED product, as a separate
          // k_split_rows pass over the fp32 result would give it - every chain fusion then stays bit-identical to its unfused form)
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
          *reinterpret_cast<uint2*>(a.o_hi + orow * a.ldop + h * DV + dv) = hh;
          *reinterpret_cast<uint2*>(a.o_lo + orow * a.ldop + h * DV + dv) = ll;
        }
      }
  } else {
    float* op = a.o + orow * a.ldo + h * DV;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int dv = t * 32 + mfma32_row(e, half);
        if (dv < DV) op[dv] = o[t][e] * inv;
      }
  }
#undef Kp
}

template <int D, int DV>
int launch_t(const AttnArgs& a, hipStream_t st) {
  if (D <= 96 && a.Lq >= 256) {   // 8 waves x 256 queries (round 2: encoder 9.04 -> 8.82 ms/frame; the DS2_ATTN_HW8 switch is gone)
    dim3 grid(cdiv(a.Lq, 256), a.heads, a.batch);
    hipLaunchKernelGGL((k_attention_bf16x3<D, DV, (D <= 96 ? 8 : 4)>), grid, dim3(D <= 96 ? 512 : 256), 0, st, a);
  } else {
    dim3 grid(cdiv(a.Lq, 128), a.heads, a.batch);
    hipLaunchKernelGGL((k_attention_bf16x3<D, DV, 4>), grid, dim3(256), 0, st, a);
  }
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

}  // namespace

#ifdef DS2_ATT_TRACE
extern "C" int ds2_debug_att_trace(unsigned long long* out) {   // [2][512] host buffer
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_att_trace), sizeof(unsigned long long) * 2 * 512) == hipSuccess ? 0 : 1;
}
#endif

// Returns DS2_ERR_UNSUPPORTED (without setting an error) for head dims that have no bf16x3 kernel yet;
// the caller then uses the exact-fp32 kernel.
int launch_attention_bf16x3(const AttnArgs& a, hipStream_t st) {
  DS2_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0, "attention: row strides must be multiples of 4");
  if (a.D == 256 && a.DV == 64 && a.heads == 1) return launch_t<256, 64>(a, st);
#define DS2_CASE(d) \
  if (a.D == d && a.DV == d) return launch_t<d, d>(a, st);
  DS2_CASE(96) DS2_CASE(72) DS2_CASE(56) DS2_CASE(32) DS2_CASE(16)
#undef DS2_CASE
  if (a.D == 256 && a.DV == 256 && a.heads == 1) {
    // DV = 256 does not fit the register file next to the 128 Q registers: four DV=64 column passes
    // (scores recomputed per pass; self-attention is 1/7 of the cross-attention work).
    for (int c = 0; c < 4; ++c) {
      AttnArgs p = a;
      p.DV = 64;
      p.v = a.v + c * 64;
      p.o = a.o + c * 64;
      int rc = launch_t<256, 64>(p, st);
      if (rc != DS2_OK) return rc;
    }
    return DS2_OK;
  }
  return DS2_ERR_UNSUPPORTED;
}
