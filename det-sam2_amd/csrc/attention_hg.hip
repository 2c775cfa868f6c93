// Hiera GLOBAL attention (hieradet.py:46-90 with window_size == 0: every token of an image attends to every token, per head;
// 3 of hiera_l's 48 blocks, 4096 x 4096 tokens, 8 heads of 72) in bf16x3 arithmetic on the structure of the memory attention
// kernel (attention_w8.hip): operands PRE-SPLIT once per (image, head) into bf16 planes - K rows padded to a multiple of 32
// columns, V^T in 32-key tiles with the keys permuted into the accumulator's row order - and a workgroup of 8 waves x 2 query
// groups x 16 queries that stages each 32-key tile with plain 16-byte copies, prefetched into registers under the MFMAs of the
// previous tile.  The general kernel (attention_bf16x3.hip) fetches fp32 K / V, splits and stages them inside the key loop of
// EVERY 256-query workgroup (16 times per image and head here) with all eight waves in the same phase: its matrix pipe is busy
// 23 % of the time (98 algorithmic TFLOP/s on this shape).
//
// Arithmetic: as attention_bf16x3.hip - every product a.b = a_lo.b_hi + a_hi.b_lo + a_hi.b_hi on v_mfma_f32_16x16x32_bf16, fp32
// accumulation, online softmax in fp32 (scores carry scale * log2 e through Q); only the summation order inside a dot product
// differs (k-steps of 32 instead of 16).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BKEYS = 32;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                       bf16x8& p0, bf16x8& p1) {
  uint4 h, l;
  h.x = cvt_pk_bf16(v0, v1); h.y = cvt_pk_bf16(v2, v3);
  h.z = cvt_pk_bf16(v4, v5); h.w = cvt_pk_bf16(v6, v7);
  l.x = cvt_pk_bf16(v0 - bf_lo(h.x), v1 - bf_hi(h.x));
  l.y = cvt_pk_bf16(v2 - bf_lo(h.y), v3 - bf_hi(h.y));
  l.z = cvt_pk_bf16(v4 - bf_lo(h.z), v5 - bf_hi(h.z));
  l.w = cvt_pk_bf16(v6 - bf_lo(h.w), v7 - bf_hi(h.w));
  p0 = __builtin_bit_cast(bf16x8, h);
  p1 = __builtin_bit_cast(bf16x8, l);
}
// the same split written with vector conversions (v_cvt_pk_bf16_f32 / shifts): instructions the scheduler can classify and place
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2v(float a, float b, bf16x2& h, bf16x2& l) {
  h = __builtin_convertvector((f32x2{a, b}), bf16x2);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  l = __builtin_convertvector((f32x2{a - hf[0], b - hf[1]}), bf16x2);
}
__device__ __forceinline__ void split8v(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                        bf16x8& p0, bf16x8& p1) {
  bf16x2 h0, h1, h2, h3, l0, l1, l2, l3;
  split2v(v0, v1, h0, l0); split2v(v2, v3, h1, l1); split2v(v4, v5, h2, l2); split2v(v6, v7, h3, l3);
  p0 = bf16x8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
  p1 = bf16x8{l0[0], l0[1], l1[0], l1[1], l2[0], l2[1], l3[0], l3[1]};
}
__device__ __forceinline__ float xmax16(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xmax32(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// ---- pre-split of K: [batch * heads][tile][plane hi, lo][32 keys][DQ + 16 bf16] - the LDS image of a key tile, row pitch included,
// so that staging a tile is a linear copy (LDS-DMA); columns DH .. DQ zero.  One thread per (row, head, 8-column chunk).
template <int DH, int DQ>
__global__ __launch_bounds__(256) void k_hg_split_k(const float* __restrict__ k, int ldk, int batch, int heads, int L,
                                                    uint4* __restrict__ hi) {
  constexpr int NCH = DQ / 8;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)batch * L * heads * NCH) return;
  const int ch = (int)(i % NCH);
  size_t r = i / NCH;
  const int h = (int)(r % heads);
  r /= heads;                                   // r = b * L + key
  const int b = (int)(r / L), key = (int)(r - (size_t)b * L);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (ch * 8 < DH) {                            // DH % 8 == 0: a chunk is all data or all pad
    const float4 a = *reinterpret_cast<const float4*>(k + r * ldk + h * DH + ch * 8);
    const float4 c = *reinterpret_cast<const float4*>(k + r * ldk + h * DH + ch * 8 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
  }
  bf16x8 p0, p1;
  split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], p0, p1);
  // tile image [plane][32 rows][NCH + 2 uint4]: exactly what the attention kernel keeps in LDS (2 pad chunks per row, never read)
  const int tile = key >> 5, row = key & 31, ntile = L / 32;
  const size_t o = ((((size_t)b * heads + h) * ntile + tile) * 2) * (32 * (NCH + 2)) + (size_t)row * (NCH + 2) + ch;
  hi[o] = __builtin_bit_cast(uint4, p0);
  hi[o + 32 * (NCH + 2)] = __builtin_bit_cast(uint4, p1);
}

// ---- pre-split of V: vt [batch * heads][tile][plane][DVP rows][32 positions + 16 pad] bf16 (the LDS image, as for K); position of key (0..31) as in attention_w8
// (vt_pos16: the accumulator's own row order), rows >= DH zero.  One thread per (tile, dv row): 32 key loads (coalesced across
// the lanes, which differ in dv), two 64-byte stores per plane row.
template <int DH, int DVP>
__global__ __launch_bounds__(256) void k_hg_split_vt(const float* __restrict__ v, int ldv, int batch, int heads, int L,
                                                     unsigned short* __restrict__ vt) {
  const int ntile = L / 32;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)batch * heads * ntile * DVP) return;
  const int dv = (int)(i % DVP);
  const size_t bt = i / DVP;                    // (b * heads + h) * ntile + tile
  const int tile = (int)(bt % ntile);
  const size_t bh = bt / ntile;
  const int h = (int)(bh % heads), b = (int)(bh / heads);
  const float* src = v + ((size_t)b * L + (size_t)tile * 32) * ldv + h * DH + dv;
  unsigned hh[16], ll[16];
#pragma unroll
  for (int key = 0; key < 32; key += 2) {
    const float x0 = dv < DH ? src[(size_t)key * ldv] : 0.f;
    const float x1 = dv < DH ? src[(size_t)(key + 1) * ldv] : 0.f;
    const unsigned hv = cvt_pk_bf16(x0, x1);
    const int pos = 8 * ((key >> 2) & 3) + (key & 3) + 4 * (key >> 4);
    hh[pos >> 1] = hv;
    ll[pos >> 1] = cvt_pk_bf16(x0 - bf_lo(hv), x1 - bf_hi(hv));
  }
  // tile image [plane][DVP rows][48 bf16]: row pitch 96 bytes as in LDS (16 pad positions per row, never read)
  uint4* oh = reinterpret_cast<uint4*>(vt + bt * 2 * (48 * DVP) + (size_t)dv * 48);
  uint4* ol = reinterpret_cast<uint4*>(vt + bt * 2 * (48 * DVP) + 48 * DVP + (size_t)dv * 48);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    oh[q4] = make_uint4(hh[4 * q4], hh[4 * q4 + 1], hh[4 * q4 + 2], hh[4 * q4 + 3]);
    ol[q4] = make_uint4(ll[4 * q4], ll[4 * q4 + 1], ll[4 * q4 + 2], ll[4 * q4 + 3]);
  }
}

struct HgArgs {
  const float* q; int ldq;
  const uint4* k_hi; const uint4* vt;   // tile images (hi and lo plane of a tile adjacent)
  float* o; int ldo;
  unsigned short *o_hi, *o_lo; int ldop;
  int batch, heads, Lq, Lk;
  float scale;
  int o_mx;
};

// DH = head dim (multiple of 8), DQ = DH rounded up to 32 (score k-steps), DVP = DH rounded up to 16 (output row blocks)
template <int DH, int DQ, int DVP>
__global__ __launch_bounds__(512, 2) void k_attention_hg(HgArgs a) {
  constexpr int QG = 2, BQ = 256, KS = DQ / 32, NT = DVP / 16;
  constexpr int KROWB = DQ * 2 + 32, KPLANE = BKEYS * KROWB;   // pitch 16 * 14 (DQ = 96) / 16 * 10 (64): conflict-free 16-lane groups
  constexpr int VROWB = 96, VPLANE = DVP * VROWB;
  constexpr int KU4 = DQ / 8;                                  // uint4 per K row and plane
  constexpr int NK = 2 * BKEYS * KU4, NKLD = (NK + 511) / 512; // staged uint4 per tile: K (2 planes), V^T (2 planes x DVP rows x 4)
  constexpr int NV = 2 * DVP * 4, NVLD = (NV + 511) / 512;
  constexpr int QSLD = DQ + 1, QBYTES = 32 * QSLD * 4;
  static_assert(2 * KPLANE >= QBYTES, "the Q staging area lives in one K buffer");
  // one LDS array, addressed by byte offsets: K image of buffer b at b * 2 KPLANE (planes hi, lo), V^T images behind them
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * KPLANE + 4 * VPLANE];
  constexpr int VOFF = 4 * KPLANE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, grp = lane >> 4;
  const int nqb = a.Lq / BQ, nblk = a.batch * a.heads * nqb;
  int bid = blockIdx.x;
  {   // the query blocks of one (image, head) on one XCD: its K / V planes stay in one L2 (bijective for any block count)
    const int xcd = bid % 8, qq = nblk / 8, rr = nblk % 8;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + bid / 8;
  }
  const int bh = bid / nqb, q0i = (bid % nqb) * BQ;
  const int b = bh / a.heads, h = bh % a.heads;
  const float sc = a.scale * 1.44269504088896340736f;

  // ---- Q: rounds of 32 rows through LDS (fp32, scaled, pad columns zero); each wave picks up its 2 x 16 rows
  bf16x8 q0[QG][KS], q1[QG][KS];
  {
    float* Qs = reinterpret_cast<float*>(lds);
    for (int r4 = 0; r4 < BQ / 32; ++r4) {
      for (int idx = tid; idx < 32 * (DQ / 4); idx += 512) {
        const int r = idx / (DQ / 4), c4 = idx - r * (DQ / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 * 4 < DH) v = *reinterpret_cast<const float4*>(a.q + ((size_t)b * a.Lq + q0i + r4 * 32 + r) * a.ldq + h * DH + c4 * 4);
        float* dst = Qs + r * QSLD + c4 * 4;
        dst[0] = v.x * sc; dst[1] = v.y * sc; dst[2] = v.z * sc; dst[3] = v.w * sc;
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        const int row = wave * 16 * QG + g * 16;
        if (row / 32 == r4) {
          const float* qrow = Qs + ((row & 31) + l15) * QSLD + grp * 8;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const float* qr = qrow + ks * 32;
            split8(qr[0], qr[1], qr[2], qr[3], qr[4], qr[5], qr[6], qr[7], q0[g][ks], q1[g][ks]);
          }
        }
      }
      __syncthreads();
    }
  }

  f32x4 o[QG][NT];
  float m_run[QG], l_run[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    m_run[g] = -INFINITY;
    l_run[g] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) o[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int nkt = a.Lk / BKEYS;
  // Staging = LDS-DMA of the tile images (HBM holds them in LDS layout): a wave issues its 2 (3) pieces of 1 KiB
  // (lane -> 16 bytes) of the K image (KIMG bytes) and of the V^T image; a piece index past the image repeats the last piece.
  // No staging registers, no ds_write; the copies run under the whole step and are waited for (vmcnt) before its barrier.
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr int KIMG = 2 * KPLANE, VIMG = 2 * VPLANE, KPC = (KIMG + 1023) / 1024, VPC = (VIMG + 1023) / 1024;
  constexpr int KPW = (KPC + 7) / 8, VPW = (VPC + 7) / 8;      // pieces per wave
  static_assert(KIMG % 1024 == 0 && VIMG % 1024 == 0, "tile images: whole KiB pieces");
  const char* kimg = reinterpret_cast<const char*>(a.k_hi) + (size_t)bh * nkt * KIMG;
  const char* vimg = reinterpret_cast<const char*>(a.vt) + (size_t)bh * nkt * VIMG;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
#define HG_DMA_K(KT, BUF)                                                     \
  _Pragma("unroll") for (int j = 0; j < KPW; ++j) {                           \
    const int pc_ = wv * KPW + j < KPC ? wv * KPW + j : KPC - 1;              \
    __builtin_amdgcn_global_load_lds(kimg + (size_t)(KT) * KIMG + pc_ * 1024 + lane * 16, (lds_ptr)(lds + (BUF) * KIMG + pc_ * 1024), 16, 0, 0); \
  }
#define HG_DMA_V(KT, BUF)                                                     \
  _Pragma("unroll") for (int j = 0; j < VPW; ++j) {                           \
    const int pc_ = wv * VPW + j < VPC ? wv * VPW + j : VPC - 1;              \
    __builtin_amdgcn_global_load_lds(vimg + (size_t)(KT) * VIMG + pc_ * 1024 + lane * 16, (lds_ptr)(lds + VOFF + (BUF) * VIMG + pc_ * 1024), 16, 0, 0); \
  }

  f32x4 sa0[QG], sa1[QG], sb0[QG], sb1[QG];   // two score sets: tile t (in the softmax) and tile t + 1 (being accumulated)
  // S^T = K Q^T for the two 16-key blocks of the tile in K buffer kb (prologue only; the steady state is step())
  auto scores = [&](int kb, f32x4 (&s0)[QG], f32x4 (&s1)[QG]) {
#pragma unroll
    for (int g = 0; g < QG; ++g) { s0[g] = f32x4{0.f, 0.f, 0.f, 0.f}; s1[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const unsigned char* kp0 = lds + kb * (2 * KPLANE) + l15 * KROWB + grp * 16;
    const unsigned char* kp1 = kp0 + KPLANE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 a00 = *reinterpret_cast<const bf16x8*>(kp0 + ks * 64);
      const bf16x8 a10 = *reinterpret_cast<const bf16x8*>(kp0 + 16 * KROWB + ks * 64);
      const bf16x8 a01 = *reinterpret_cast<const bf16x8*>(kp1 + ks * 64);
      const bf16x8 a11 = *reinterpret_cast<const bf16x8*>(kp1 + 16 * KROWB + ks * 64);
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        s0[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01, q0[g][ks], s0[g], 0, 0, 0);
        s0[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a00, q1[g][ks], s0[g], 0, 0, 0);
        s0[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a00, q0[g][ks], s0[g], 0, 0, 0);
        s1[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a11, q0[g][ks], s1[g], 0, 0, 0);
        s1[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, q1[g][ks], s1[g], 0, 0, 0);
        s1[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, q0[g][ks], s1[g], 0, 0, 0);
      }
    }
  };
  // ---- one step = 2 KS pinned segments (sched_barrier(0) between them; hipcc left alone issues the 12 KS score MFMAs back to back
  // and then ~150 VALU instructions of the softmax in a row, and one sched_group_barrier pipeline over the whole block defeats the
  // greedy solver).  Segment (ks, key block): the fragment reads of the segment after next, the 6 score MFMAs of tile t+1 for this
  // k-step and key block (3 product terms x 2 query groups), and a share of the softmax of tile t in their shadow.  Empty volatile
  // statements fence each share into its segment (sched_barrier does not order register-only arithmetic during selection).
  float sm_m[QG], sm_a[QG], sm_p[QG][8];
  bf16x8 sm_pb0[QG], sm_pb1[QG];
  auto sm_part = [&](auto part_tag, auto g_tag, f32x4 (&s0)[QG], f32x4 (&s1)[QG]) {
    constexpr int PART = decltype(part_tag)::value, g = decltype(g_tag)::value;
    if constexpr (PART == 0) {          // running maximum, rescale factor, first four weights
      asm volatile("" : "+v"(m_run[g]));
      float tmax = fmaxf(fmaxf(fmaxf(s0[g][0], s0[g][1]), fmaxf(s0[g][2], s0[g][3])),
                         fmaxf(fmaxf(s1[g][0], s1[g][1]), fmaxf(s1[g][2], s1[g][3])));
      tmax = xmax16(tmax);
      tmax = xmax32(tmax);
      sm_m[g] = fmaxf(m_run[g], tmax);
      sm_a[g] = __builtin_amdgcn_exp2f(m_run[g] - sm_m[g]);
      m_run[g] = sm_m[g];
#pragma unroll
      for (int r = 0; r < 4; ++r) sm_p[g][r] = __builtin_amdgcn_exp2f(s0[g][r] - sm_m[g]);
      asm volatile("" : "+v"(sm_m[g]), "+v"(sm_a[g]), "+v"(sm_p[g][0]), "+v"(sm_p[g][1]), "+v"(sm_p[g][2]), "+v"(sm_p[g][3]));
    } else if constexpr (PART == 1) {   // the other four weights, the row sum
      asm volatile("" : "+v"(sm_m[g]));
#pragma unroll
      for (int r = 0; r < 4; ++r) sm_p[g][4 + r] = __builtin_amdgcn_exp2f(s1[g][r] - sm_m[g]);
      l_run[g] = l_run[g] * sm_a[g] + (((sm_p[g][0] + sm_p[g][1]) + (sm_p[g][2] + sm_p[g][3])) + ((sm_p[g][4] + sm_p[g][5]) + (sm_p[g][6] + sm_p[g][7])));
      asm volatile("" : "+v"(l_run[g]), "+v"(sm_p[g][4]), "+v"(sm_p[g][5]), "+v"(sm_p[g][6]), "+v"(sm_p[g][7]));
    } else {                            // split of P into its two bf16 planes
      asm volatile("" : "+v"(sm_p[g][0]));
      split8v(sm_p[g][0], sm_p[g][1], sm_p[g][2], sm_p[g][3], sm_p[g][4], sm_p[g][5], sm_p[g][6], sm_p[g][7], sm_pb0[g], sm_pb1[g]);
      asm volatile("" : "+v"(sm_pb0[g]), "+v"(sm_pb1[g]));
    }
  };
  auto step = [&](int kb, int vbuf, f32x4 (&c0)[QG], f32x4 (&c1)[QG], f32x4 (&n0)[QG], f32x4 (&n1)[QG]) {
    constexpr int NSEG = 2 * KS;
    const unsigned char* kp0 = lds + kb * (2 * KPLANE) + l15 * KROWB + grp * 16;
    bf16x8 fh[3], fl[3];                 // fragment ring: segment s reads for segment s + 2
    bf16x8 vf0[NT], vf1[NT];             // the tile's V^T fragments, read in the last two segments (under the score MFMAs)
    const unsigned char* vp0 = lds + VOFF + vbuf * (2 * VPLANE) + l15 * VROWB + grp * 16;
    auto frag = [&](int sgi, int slot) {
      const int ks = sgi >> 1, blk = sgi & 1;
      fh[slot] = *reinterpret_cast<const bf16x8*>(kp0 + blk * 16 * KROWB + ks * 64);
      fl[slot] = *reinterpret_cast<const bf16x8*>(kp0 + KPLANE + blk * 16 * KROWB + ks * 64);
    };
    frag(0, 0);
    frag(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto seg = [&](auto s_tag) {
      constexpr int sgi = decltype(s_tag)::value, ks = sgi >> 1, blk = sgi & 1;
      constexpr int NTA = (NT + 1) / 2, vt0 = sgi + 2 == NSEG ? 0 : NTA, vt1 = sgi + 2 == NSEG ? NTA : NT;
      if constexpr (sgi + 2 < NSEG) {
        frag(sgi + 2, (sgi + 2) % 3);
      } else {
#pragma unroll
        for (int t = vt0; t < vt1; ++t) {
          vf0[t] = *reinterpret_cast<const bf16x8*>(vp0 + t * 16 * VROWB);
          vf1[t] = *reinterpret_cast<const bf16x8*>(vp0 + VPLANE + t * 16 * VROWB);
        }
      }
      // (term-major: consecutive MFMAs never target the same accumulator)
#pragma unroll
      for (int g = 0; g < QG; ++g) (blk ? n1[g] : n0[g]) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[sgi % 3], q0[g][ks], ks == 0 ? zero4 : (blk ? n1[g] : n0[g]), 0, 0, 0);
#pragma unroll
      for (int g = 0; g < QG; ++g) (blk ? n1[g] : n0[g]) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[sgi % 3], q1[g][ks], blk ? n1[g] : n0[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < QG; ++g) (blk ? n1[g] : n0[g]) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[sgi % 3], q0[g][ks], blk ? n1[g] : n0[g], 0, 0, 0);
      // softmax shares: segments [0, NSEG/2) serve query group 0, the rest group 1; three parts spread over NSEG/2 segments
      constexpr int sg = sgi / (NSEG / 2), loc = sgi % (NSEG / 2), per = NSEG / 2;
      if constexpr (per >= 3) {
        if constexpr (loc < 3) sm_part(std::integral_constant<int, loc>{}, std::integral_constant<int, sg>{}, c0, c1);
      } else {   // KS = 2: two segments per group
        if constexpr (loc == 0) {
          sm_part(std::integral_constant<int, 0>{}, std::integral_constant<int, sg>{}, c0, c1);
          sm_part(std::integral_constant<int, 1>{}, std::integral_constant<int, sg>{}, c0, c1);
        } else {
          sm_part(std::integral_constant<int, 2>{}, std::integral_constant<int, sg>{}, c0, c1);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, sgi + 2 < NSEG ? 2 : 2 * (vt1 - vt0), 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    seg(std::integral_constant<int, 0>{}); seg(std::integral_constant<int, 1>{}); seg(std::integral_constant<int, 2>{}); seg(std::integral_constant<int, 3>{});
    if constexpr (NSEG > 4) { seg(std::integral_constant<int, 4>{}); seg(std::integral_constant<int, 5>{}); }
    // ---- O^T += V^T P^T (tile t)
    bool ch = false;
#pragma unroll
    for (int g = 0; g < QG; ++g) ch |= sm_a[g] != 1.f;
    if (__any(ch)) {   // (exact: alpha == 1 leaves o unchanged; after the first tiles the common case)
#pragma unroll
      for (int g = 0; g < QG; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t) o[g][t] *= sm_a[g];
    }
    // (term-major over all row blocks: 2 NT independent accumulators between two MFMAs on the same one)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < QG; ++g) o[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1[t], sm_pb0[g], o[g][t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < QG; ++g) o[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0[t], sm_pb1[g], o[g][t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < QG; ++g) o[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0[t], sm_pb0[g], o[g][t], 0, 0, 0);
  };

  HG_DMA_K(0, 0)
  HG_DMA_V(0, 0)
  HG_DMA_K(nkt > 1 ? 1 : 0, 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // one step: [DMA of K(kt+2), V(kt+1) issued] [scores of tile kt+1] [softmax + P.V of tile kt] [DMA landed] [barrier]; the scores
  // of a tile past the end run on the last K buffer and are ignored (no branch inside the step)
#define HG_STEP(KT, C0, C1, N0, N1)                                           \
  {                                                                           \
    const int kt_s = (KT);                                                    \
    HG_DMA_K(kt_s + 2 < nkt ? kt_s + 2 : nkt - 1, kt_s & 1)                   \
    HG_DMA_V(kt_s + 1 < nkt ? kt_s + 1 : nkt - 1, (kt_s + 1) & 1)             \
    step((kt_s + 1) & 1, kt_s & 1, C0, C1, N0, N1);                           \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          \
    __syncthreads();                                                          \
  }
  scores(0, sa0, sa1);
  __syncthreads();   // every wave has read K(0) before step 0's DMA overwrites it with K(2)
  for (int kt = 0; kt < nkt; kt += 2) {
    HG_STEP(kt, sa0, sa1, sb0, sb1)
    if (kt + 1 < nkt) HG_STEP(kt + 1, sb0, sb1, sa0, sa1)
  }

#pragma unroll
  for (int g = 0; g < QG; ++g) {
    float l_tot = l_run[g] + __shfl_xor(l_run[g], 16);
    l_tot += __shfl_xor(l_tot, 32);
    const float inv = 1.f / l_tot;
    const size_t orow = (size_t)b * a.Lq + q0i + wave * 16 * QG + g * 16 + l15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int dv = 16 * t + 4 * grp;       // this lane's 4 consecutive output columns (DH % 4 == 0: all valid or all pad)
      if (dv >= DH) continue;
      const float v0 = o[g][t][0] * inv, v1 = o[g][t][1] * inv, v2 = o[g][t][2] * inv, v3 = o[g][t][3] * inv;
      if (a.o_hi) {
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
      } else {
        *reinterpret_cast<float4*>(a.o + orow * a.ldo + h * DH + dv) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}

template <int DH>
int launch_t(const AttnArgs& a, void* k_hi, void* vt, hipStream_t st) {
  constexpr int DQ = (DH + 31) / 32 * 32, DVP = (DH + 15) / 16 * 16;
  const size_t nk = (size_t)a.batch * a.Lk * a.heads * (DQ / 8);
  hipLaunchKernelGGL((k_hg_split_k<DH, DQ>), dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, a.k, a.ldk, a.batch, a.heads, a.Lk,
                     reinterpret_cast<uint4*>(k_hi));
  DS2_CHECK_LAUNCH();
  const size_t nv = (size_t)a.batch * a.heads * (a.Lk / 32) * DVP;
  hipLaunchKernelGGL((k_hg_split_vt<DH, DVP>), dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, a.v, a.ldv, a.batch, a.heads, a.Lk,
                     reinterpret_cast<unsigned short*>(vt));
  DS2_CHECK_LAUNCH();
  HgArgs g{a.q, a.ldq, reinterpret_cast<const uint4*>(k_hi), reinterpret_cast<const uint4*>(vt),
           a.o, a.ldo, a.o_hi, a.o_lo, a.ldop, a.batch, a.heads, a.Lq, a.Lk, a.scale, a.o_mx};
  hipLaunchKernelGGL((k_attention_hg<DH, DQ, DVP>), dim3(a.batch * a.heads * (a.Lq / 256)), dim3(512), 0, st, g);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

}  // namespace

// Plain (window-less) multi-head attention with D == DV in {56, 72, 96}, Lq a multiple of 256, Lk a multiple of 32, rows 16-byte
// aligned: the Hiera global-attention blocks of every SAM 2.1 configuration.
bool attention_hg_supported(const AttnArgs& a) {
  return a.win_k == 0 && a.D == a.DV && (a.D == 56 || a.D == 72 || a.D == 96) && a.Lq % 256 == 0 && a.Lk % 32 == 0 && a.Lk >= 64 &&
         a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldo % 4 == 0 && (!a.o_hi || a.ldop % 4 == 0);
}
// bytes of the K tile images (both planes, padded rows) and of the V^T tile images for attention_hg_supported() arguments
size_t attention_hg_k_bytes(const AttnArgs& a) { return (size_t)a.batch * a.heads * a.Lk * (((a.D + 31) / 32 * 32) * 2 + 32) * 2; }
size_t attention_hg_vt_bytes(const AttnArgs& a) { return (size_t)a.batch * a.heads * (a.Lk / 32) * ((a.D + 15) / 16 * 16) * 96 * 2; }

int launch_attention_hg(const AttnArgs& a, void* k_img, void* vt_img, hipStream_t st) {
  DS2_REQUIRE(attention_hg_supported(a) && k_img && vt_img && (a.o || a.o_hi), "attention_hg: unsupported arguments");
  if (a.D == 72) return launch_t<72>(a, k_img, vt_img, st);
  if (a.D == 96) return launch_t<96>(a, k_img, vt_img, st);
  return launch_t<56>(a, k_img, vt_img, st);
}
