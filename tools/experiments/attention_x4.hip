// Memory CROSS-attention of mode bf16x3k, round-4 form: 4 waves x 64 queries, ONE wave per SIMD, v_mfma_f32_32x32x16_f16.
//
// (RoPEAttention.forward of the memory attention's cross_attn_image, sam2/modeling/sam/transformer.py:312-363, in the
// restructured form of DESIGN.md section 4: softmax(Q K^T) M - the values are the raw 64-d memory, v_proj is applied after.)
//
// Why a second kernel (attention_w8.hip keeps the bf16x3 mode, the self-attention, ragged grids and few objects): the 8-wave
// kernel runs 16 queries per MFMA column block and two waves per SIMD; per 32-key tile and SIMD it issues 80 16x16x32 MFMAs
// (1 280 matrix-pipe cycles), reads 40 KiB of K / V^T fragments (every fragment is fetched by all 8 waves) and runs its
// softmax in two waves that arbitrate for the VALU - measured 2 670 cycles per tile (r03: 0.42 of the bf16 peak, MFMA busy
// 50 %).  Here
//   * a wave owns 64 queries (two 32-column blocks of the 32x32 MFMA) and the WHOLE 512-entry register file of its SIMD: the
//     queries stay in registers as the B operand for the entire key loop (128 registers of fp16 fragments), so a K fragment
//     read from LDS feeds two MFMAs of 32 cycles - LDS reads per matrix cycle are a quarter of the 8-wave kernel's;
//   * scores are computed TRANSPOSED, S^T = K Q^T: a lane owns one query column and 16 of the 32 keys of a tile (its partner
//     lane ^ 32 the other 16), so the softmax statistics are per-lane scalars + one v_permlane32_swap, and P^T - converted to
//     fp16 in place - IS the B operand of O^T += V^T P^T (V^T is stored key-permuted to the accumulator's row order by
//     k_vt_pack32): no LDS or cross-lane traffic for P;
//   * K / V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, XOR-swizzled source addresses so
//     that the un-padded LDS images are conflict-free for ds_read_b128) into 4-slot rings, K three tiles and V^T two tiles
//     ahead of their use; ONE barrier per tile, counted vmcnt (the copies of the current iteration stay in flight across it);
//   * the step is one basic block: the 32 score MFMAs of tile t+1 carry the softmax VALU work of tile t in their shadow
//     (sched_group_barrier pipeline), then the 8 P.V MFMAs of tile t.
// All single-plane operands are IEEE fp16 (DS2_ATTN_K_F16, kernels.h); softmax, running maximum / sum and all accumulation fp32.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int D = 256, DV = 64, BK = 32, KSTEPS = D / 16, RING = 4;
constexpr int KT_BYTES = BK * D * 2;      // 16 KiB: one K tile (32 rows of 512 B)
constexpr int VT_BYTES = DV * BK * 2;     // 4 KiB: one V^T tile (64 rows of 64 B)
constexpr int V_OFF = RING * KT_BYTES;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {   // v_cvt_pk_f16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{a, b}), f16x2));
}
__device__ __forceinline__ f16x8 pack8h(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  const f16x2 a = __builtin_convertvector((f32x2{v0, v1}), f16x2), b = __builtin_convertvector((f32x2{v2, v3}), f16x2);
  const f16x2 c = __builtin_convertvector((f32x2{v4, v5}), f16x2), d = __builtin_convertvector((f32x2{v6, v7}), f16x2);
  return f16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}
__device__ __forceinline__ float xmax32(float x) {   // max over lanes {l, l ^ 32}
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// key (0..31 inside a tile) -> slot of its value in a V^T row.  The 32x32 accumulator of a lane in half h holds, in register
// r, key (r & 3) + 8 (r >> 2) + 4 h; P.V step s (16 keys of MFMA depth) takes registers 8 s .. 8 s + 7 as the B operand, whose
// k index is 8 h + j.  So slot 16 s + 8 h + j <-> key 16 s + 8 (j >> 2) + 4 h + (j & 3).
__host__ __device__ inline int vt_pos32(int key) { return (key & 16) + 8 * ((key >> 2) & 1) + 4 * ((key >> 3) & 1) + (key & 3); }

// vt[b][tile][dv 0..63][slot 0..31] fp16.  One thread per (tile, dv) row: 32 key loads (coalesced across the lanes, which
// differ in dv) and one 64-byte row store.  Keys beyond L are zero.
__global__ __launch_bounds__(256) void k_vt_pack32(const float* __restrict__ v, int ldv, int batch, int L,
                                                   unsigned short* __restrict__ vt) {
  const int ntile = (L + 31) / 32;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)batch * ntile * DV) return;
  const int dv = (int)(i % DV);
  const size_t bt = i / DV;
  const int tile = (int)(bt % ntile), b = (int)(bt / ntile);
  const float* src = v + ((size_t)b * L + (size_t)tile * 32) * ldv + dv;
  const int nvalid = L - tile * 32;
  unsigned h[16];
#pragma unroll
  for (int key = 0; key < 32; key += 2) {    // keys (key, key + 1) sit at adjacent slots (pos, pos + 1), pos even
    const float x0 = key < nvalid ? src[(size_t)key * ldv] : 0.f;
    const float x1 = key + 1 < nvalid ? src[(size_t)(key + 1) * ldv] : 0.f;
    h[vt_pos32(key) >> 1] = cvt_pk_f16(x0, x1);
  }
  uint4* o = reinterpret_cast<uint4*>(vt + (bt * DV + dv) * 32);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) o[q4] = make_uint4(h[4 * q4], h[4 * q4 + 1], h[4 * q4 + 2], h[4 * q4 + 3]);
}

struct X4Args {
  const float* q; int ldq;
  const char* k;              // fp16 key plane [batch * Lk][256] (k_proj + RoPE epilogue of the K = 64 GEMM)
  const char* vt;             // k_vt_pack32 tiles
  unsigned short *o_hi, *o_lo; int ldop;   // result as bf16 operand planes [batch * Lq, ldop] (consumer: the folded v/out projection)
  int batch, Lq, Lk;
  float scale;
  const float* rope_cis; int rope_grid, rope_w;
  int q_bstride;              // rows between the query blocks of consecutive batch items (0: shared queries, layer 0)
};

#ifndef DS2_X4_ILV
#define DS2_X4_ILV 1
#endif
#ifndef DS2_X4_ASM_MFMA
#define DS2_X4_ASM_MFMA 1
#endif
#ifndef DS2_X4_PINQ
#define DS2_X4_PINQ 1
#endif
#ifndef DS2_X4_VALU_PER_MFMA
#define DS2_X4_VALU_PER_MFMA 4
#endif

template <bool ROPE>
__global__ __launch_bounds__(256, 1) void k_attention_x4(X4Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[RING * (KT_BYTES + VT_BYTES)];   // 80 KiB
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nqb = a.Lq / 256, nblk = a.batch * nqb;
  int bid = blockIdx.x;
  {   // consecutive block ids (= the query blocks of one object) on one XCD: its K / V^T stay in one L2 (bijective, T1)
    const int xcd = bid % 8, qq = nblk / 8, rr = nblk % 8;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + bid / 8;
  }
  const int b = bid / nqb, q0i = (bid % nqb) * 256;
  const float sc = a.scale * 1.44269504088896340736f;
  const int nkt = (a.Lk + BK - 1) / BK;

  // ---- staging: wave w copies K pieces 4w .. 4w+3 (1 KiB = two 512-byte rows: lane -> row 2 pc + lane / 32, physical chunk
  // lane % 32 = the global chunk (lane % 32) ^ (row & 15)) and V^T piece w (16 rows of 64 bytes: lane -> row 16 w + lane / 4,
  // physical chunk lane & 3 = the global chunk (lane & 3) ^ f((row >> 2) & 3), f = [0, 3, 2, 1]).  Per-lane source offsets
  // inside a tile: piece j of this wave = (koff0 ^ (j << 5)) + 1024 j (row & 15 = c + 2 j with c's bits disjoint from 2 j).
  const char* kbase = a.k + (size_t)b * a.Lk * 512;
  const char* vbase = a.vt + (size_t)b * nkt * VT_BYTES;
  const int krow0 = 8 * wave + (lane >> 5);
  const unsigned koff0 = (unsigned)(krow0 * 512 + (((lane & 31) ^ (krow0 & 15)) << 4));
  const int vrow = 16 * wave + (lane >> 2);
  const unsigned voff = (unsigned)(vrow * 64 + (((lane & 3) ^ ((4 - ((vrow >> 2) & 3)) & 3)) << 4));
  auto dma_k = [&](int kt, int slot) {   // (tiles past the end: the last tile again - never read)
    kt = kt < nkt ? kt : nkt - 1;
    const char* src = kbase + (size_t)kt * KT_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds(src + ((koff0 ^ (unsigned)(j << 5)) + (unsigned)(j * 1024)),
                                       (lds_ptr)(lds + slot * KT_BYTES + (4 * wave + j) * 1024), 16, 0, 0);
  };
  auto dma_v = [&](int kt, int slot) {
    kt = kt < nkt ? kt : nkt - 1;
    __builtin_amdgcn_global_load_lds(vbase + (size_t)kt * VT_BYTES + voff, (lds_ptr)(lds + V_OFF + slot * VT_BYTES + wave * 1024), 16, 0, 0);
  };
  dma_k(0, 0); dma_v(0, 0);
  dma_k(1, 1); dma_v(1, 1);
  dma_k(2, 2);

  // ---- Q: this wave's 64 rows as B fragments [q block][k step]: lane (q = l31, d = 16 ks + 8 half .. + 7), rotated (RoPE),
  // scaled by scale * log2(e), fp16.  Loaded under the first tiles' DMA.
  f16x8 qf[2][KSTEPS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = q0i + wave * 64 + qb * 32 + l31;
    const float* src = a.q + ((size_t)b * a.q_bstride + row) * a.ldq + half * 8;
    // RoPE table rows: pairs < 64 depend on x = t % w only, the others on y (row t - t % w); without the compact form (rope_w
    // == 0) both are row t.  Branch-free so that the 32 row loads of a query block leave together.
    const int t = ROPE ? row % a.rope_grid : 0;
    const int tx = a.rope_w > 0 ? t % a.rope_w : t, ty = a.rope_w > 0 ? t - t % a.rope_w : t;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      float4 v0 = *reinterpret_cast<const float4*>(src + ks * 16);
      float4 v1 = *reinterpret_cast<const float4*>(src + ks * 16 + 4);
      if constexpr (ROPE) {   // complex pairs (d, d + 1): the same expression as k_rope / the 8-wave kernel's query load
        const int pair0 = ks * 8 + half * 4;
        const int tt = ks < KSTEPS / 2 ? tx : ty;
        const float4 c0 = *reinterpret_cast<const float4*>(a.rope_cis + ((size_t)tt * 128 + pair0) * 2);
        const float4 c1 = *reinterpret_cast<const float4*>(a.rope_cis + ((size_t)tt * 128 + pair0 + 2) * 2);
        v0 = make_float4(v0.x * c0.x - v0.y * c0.y, v0.x * c0.y + v0.y * c0.x, v0.z * c0.z - v0.w * c0.w, v0.z * c0.w + v0.w * c0.z);
        v1 = make_float4(v1.x * c1.x - v1.y * c1.y, v1.x * c1.y + v1.y * c1.x, v1.z * c1.z - v1.w * c1.w, v1.z * c1.w + v1.w * c1.z);
      }
      qf[qb][ks] = pack8h(v0.x * sc, v0.y * sc, v0.z * sc, v0.w * sc, v1.x * sc, v1.y * sc, v1.z * sc, v1.w * sc);
    }
  }

  f32x16 o[2][2];   // [dv block][q block]: O^T rows dv = 32 dvb + mfma32_row(r, half), column q = l31
  float m_run[2], l_run[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int dvb = 0; dvb < 2; ++dvb)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dvb][qb][e] = 0.f;
  }

  // lane constants of the fragment reads: byte offset of k-step ks inside this lane's K row / of P.V step st inside its V^T
  // rows (swizzled chunks); the ring slot is a compile-time immediate of the ds_read (the key loop is unrolled over the 4 slots)
  unsigned kro[KSTEPS], vro[2][2];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) kro[ks] = (unsigned)(l31 * 512 + (((ks * 2 + half) ^ (l31 & 15)) << 4));
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int dvb = 0; dvb < 2; ++dvb) {
      const int row = dvb * 32 + l31;
      vro[st][dvb] = (unsigned)(V_OFF + row * 64 + (((st * 2 + half) ^ ((4 - ((row >> 2) & 3)) & 3)) << 4));
    }

  // ---- one step, hand-placed: the 16 k-steps of the scores of tile t+1 (2 MFMAs of 32 cycles each: one per query block) are
  // 16 SLOTS fenced by sched_barrier(0); slot ks also issues the K fragment read of k-step ks+2 and carries one CHUNK of the
  // softmax of tile t (whose scores `cur` were finished one iteration ago) in the shadow of its MFMAs:
  //   chunks 0-1  running maximum (30 v_max + lane swap), rescale factor
  //   chunks 2-9  p = exp2(s - m): two elements of each query block per chunk
  //   chunks 10-13 row sums, fp16 packing of P^T
  //   chunk 14    running sum, V^T fragment reads
  // then the 8 P.V MFMAs.  hipcc left to itself issues read -> wait -> 2 MFMAs per k-step with ONE fragment register set and the
  // whole softmax afterwards (1.75 ms per launch against the 8-wave kernel's 1.2).
  f16x8 pf[2][2];
  float alpha[2], p[2][16], m_new[2], psum[2];
  f16x8 vfr[2][2];
  auto chunk = [&](auto ks_tag, f32x16 (&s)[2], auto vslot_tag) {
    constexpr int vslot = decltype(vslot_tag)::value;
    constexpr int ks = decltype(ks_tag)::value;
    if constexpr (ks == 0) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
        m_new[qb] = fmaxf(fmaxf(fmaxf(s[qb][0], s[qb][1]), fmaxf(s[qb][2], s[qb][3])), fmaxf(fmaxf(s[qb][4], s[qb][5]), fmaxf(s[qb][6], s[qb][7])));
    } else if constexpr (ks == 1) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float tmax = fmaxf(fmaxf(fmaxf(s[qb][8], s[qb][9]), fmaxf(s[qb][10], s[qb][11])), fmaxf(fmaxf(s[qb][12], s[qb][13]), fmaxf(s[qb][14], s[qb][15])));
        tmax = xmax32(fmaxf(tmax, m_new[qb]));
        m_new[qb] = fmaxf(m_run[qb], tmax);
        alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_new[qb]);
        m_run[qb] = m_new[qb];
      }
    } else if constexpr (ks >= 2 && ks <= 9) {
      constexpr int e0 = 2 * (ks - 2);
      // (the four subtractions first: back-to-back dependent VALU stalls a lone wave)
      const float d0 = s[0][e0] - m_new[0], d1 = s[0][e0 + 1] - m_new[0], d2 = s[1][e0] - m_new[1], d3 = s[1][e0 + 1] - m_new[1];
      p[0][e0] = __builtin_amdgcn_exp2f(d0); p[0][e0 + 1] = __builtin_amdgcn_exp2f(d1);
      p[1][e0] = __builtin_amdgcn_exp2f(d2); p[1][e0 + 1] = __builtin_amdgcn_exp2f(d3);
    } else if constexpr (ks >= 10 && ks <= 13) {
      constexpr int g = ks - 10;   // elements 4 g .. 4 g + 3
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float part = (p[qb][4 * g] + p[qb][4 * g + 1]) + (p[qb][4 * g + 2] + p[qb][4 * g + 3]);
        psum[qb] = g == 0 ? part : psum[qb] + part;
      }
      if constexpr (g == 1 || g == 3) {
        constexpr int st = g >> 1;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          pf[qb][st] = pack8h(p[qb][8 * st], p[qb][8 * st + 1], p[qb][8 * st + 2], p[qb][8 * st + 3], p[qb][8 * st + 4], p[qb][8 * st + 5],
                              p[qb][8 * st + 6], p[qb][8 * st + 7]);
      }
    } else if constexpr (ks == 14) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) l_run[qb] = l_run[qb] * alpha[qb] + psum[qb];
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int dvb = 0; dvb < 2; ++dvb)
          vfr[st][dvb] = *reinterpret_cast<const f16x8*>(lds + vro[st][dvb] + vslot * VT_BYTES);
    }
  };
  // Pins a slot's work to the slot: an empty volatile statement that (a) consumes and re-defines the score sets - the next slot's
  // MFMAs and every later chunk depend on it, nothing of theirs can be hoisted above - and (b) consumes what this slot's chunk
  // produced - nothing of it can sink below.  (sched_barrier alone does not hold register-only VALU: it is placed before the
  // scheduler runs.)  "v": the scores stay in arch VGPRs (the softmax reads them; no v_accvgpr_read copies).
#define X4_FENCE(KS, c, n)                                                                                                      \
  {                                                                                                                             \
    constexpr int ks = (KS);                                                                                                    \
    if constexpr (ks == 0) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(n[0]), "+v"(n[1]), "+v"(m_new[0]), "+v"(m_new[1]));   \
    else if constexpr (ks == 1)                                                                                                 \
      asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(n[0]), "+v"(n[1]), "+v"(m_new[0]), "+v"(m_new[1]), "+v"(alpha[0]), "+v"(alpha[1])); \
    else if constexpr (ks >= 2 && ks <= 9)                                                                                      \
      asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(n[0]), "+v"(n[1]), "+v"(p[0][ks >= 2 && ks <= 9 ? 2 * (ks - 2) : 0]),      \
                   "+v"(p[0][ks >= 2 && ks <= 9 ? 2 * (ks - 2) + 1 : 0]), "+v"(p[1][ks >= 2 && ks <= 9 ? 2 * (ks - 2) : 0]),    \
                   "+v"(p[1][ks >= 2 && ks <= 9 ? 2 * (ks - 2) + 1 : 0]));                                                      \
    else if constexpr (ks == 10 || ks == 12) asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(psum[0]), "+v"(psum[1]));           \
    else if constexpr (ks == 11 || ks == 13)                                                                                    \
      asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(psum[0]), "+v"(psum[1]), "+v"(pf[0][ks == 13]), "+v"(pf[1][ks == 13]));    \
    else if constexpr (ks == 14)                                                                                                \
      asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(l_run[0]), "+v"(l_run[1]), "+v"(vfr[0][0]), "+v"(vfr[0][1]), "+v"(vfr[1][0]), "+v"(vfr[1][1])); \
    else asm volatile("" : "+v"(n[0]), "+v"(n[1]));                                                                             \
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define X4_KREAD(KSLOT, KS) (*reinterpret_cast<const f16x8*>(lds + kro[KS] + (KSLOT) * KT_BYTES))

  // K(0), V^T(0), K(1), V^T(1) landed (K(2)'s 4 copies may still fly): visible to everybody after the barrier
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x16 sa[2], sb[2];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {   // scores of tile 0
    const f16x8 kf = X4_KREAD(0, ks);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) sa[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ks], ks == 0 ? zero16 : sa[qb], 0, 0, 0);
  }

  // iteration t (ring slots are compile-time: the loop is unrolled over t mod 4): [copies K(t+3), V^T(t+2)] [scores of tile t+1
  // with the softmax of tile t in their shadow] [P.V of tile t] [counted wait] [barrier].  Hazards (one barrier per iteration): K
  // slot (t+3) & 3 was last read by the scores of tile t-1 in iteration t-2, V^T slot (t+2) & 3 by P.V of tile t-2 in iteration
  // t-2; K(t+1) / V^T(t) read here were issued in iteration t-2 and waited for by every wave's vmcnt(5) at the end of iteration
  // t-1 (only that iteration's own 5 copies may be in flight).
  // Score MFMA in assembly: the query fragment is an "a" operand, so the 128 registers of Q are ALLOCATED in the accumulator half
  // (left to the allocator they sit in arch VGPRs, are spilled to AGPRs and come back through v_accvgpr_read before every use);
  // the accumulator is a "v" operand: the softmax reads it without copies.  Hazards: K fragment = the compiler's own ds_read
  // (it waits); accumulate chain MFMA -> MFMA needs no software wait; the first VALU reader of the result is the next
  // iteration's softmax, behind the P.V MFMAs and a barrier.
#if DS2_X4_ASM_MFMA
#define X4_MFMA_S(KS, ACC, KF, QF)                                                                  \
  if constexpr ((KS) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "v"(KF), "a"(QF)); \
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(KF), "a"(QF));
#else
#define X4_MFMA_S(KS, ACC, KF, QF) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(KF, QF, (KS) == 0 ? zero16 : ACC, 0, 0, 0);
#endif
#define X4_SLOT(KS, SL, CUR, NXT)                                                                   \
  {                                                                                                 \
    if constexpr ((KS) + 2 < KSTEPS) kf_[((KS) + 2) % 3] = X4_KREAD(((SL) + 1) & 3, ((KS) + 2 < KSTEPS ? (KS) + 2 : 0)); \
    X4_MFMA_S(KS, NXT[0], kf_[(KS) % 3], qf[0][KS])                                                 \
    X4_MFMA_S(KS, NXT[1], kf_[(KS) % 3], qf[1][KS])                                                 \
    chunk(std::integral_constant<int, (KS)>{}, CUR, std::integral_constant<int, (SL)>{});           \
    X4_FENCE(KS, CUR, NXT)                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  }
#define X4_STEP(T, SL, CUR, NXT)                                                                    \
  {                                                                                                 \
    const int t_ = (T);                                                                             \
    dma_k(t_ + 3, ((SL) + 3) & 3);                                                                  \
    dma_v(t_ + 2, ((SL) + 2) & 3);                                                                  \
    f16x8 kf_[3];                                                                                   \
    kf_[0] = X4_KREAD(((SL) + 1) & 3, 0);                                                           \
    kf_[1] = X4_KREAD(((SL) + 1) & 3, 1);                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    X4_SLOT(0, SL, CUR, NXT) X4_SLOT(1, SL, CUR, NXT) X4_SLOT(2, SL, CUR, NXT) X4_SLOT(3, SL, CUR, NXT)       \
    X4_SLOT(4, SL, CUR, NXT) X4_SLOT(5, SL, CUR, NXT) X4_SLOT(6, SL, CUR, NXT) X4_SLOT(7, SL, CUR, NXT)       \
    X4_SLOT(8, SL, CUR, NXT) X4_SLOT(9, SL, CUR, NXT) X4_SLOT(10, SL, CUR, NXT) X4_SLOT(11, SL, CUR, NXT)     \
    X4_SLOT(12, SL, CUR, NXT) X4_SLOT(13, SL, CUR, NXT) X4_SLOT(14, SL, CUR, NXT) X4_SLOT(15, SL, CUR, NXT)   \
    /* exact running-maximum rescale: rare after the first tiles (the volatile statement keeps it a real branch) */ \
    if (__builtin_expect(__any(alpha[0] != 1.f || alpha[1] != 1.f), 0)) {                           \
      asm volatile("; x4 rescale");                                                                 \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                              \
        _Pragma("unroll") for (int dvb = 0; dvb < 2; ++dvb)                                         \
          _Pragma("unroll") for (int e = 0; e < 16; ++e) o[dvb][qb][e] *= alpha[qb];                \
    }                                                                                               \
    _Pragma("unroll") for (int st = 0; st < 2; ++st)                                                \
      _Pragma("unroll") for (int dvb = 0; dvb < 2; ++dvb)                                           \
        _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                            \
          o[dvb][qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfr[st][dvb], pf[qb][st], o[dvb][qb], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                                \
    __builtin_amdgcn_s_barrier();                                                                   \
  }
  // (full tiles only - attention_x4_supported: a ragged key count keeps the 8-wave kernel)
  int t = 0;
  for (; t + 3 < nkt; t += 4) {
    X4_STEP(t, 0, sa, sb)
    X4_STEP(t + 1, 1, sb, sa)
    X4_STEP(t + 2, 2, sa, sb)
    X4_STEP(t + 3, 3, sb, sa)
  }
  if (t < nkt) {
    X4_STEP(t, 0, sa, sb)
    if (t + 1 < nkt) {
      X4_STEP(t + 1, 1, sb, sa)
      if (t + 2 < nkt) X4_STEP(t + 2, 2, sa, sb)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (surplus copies of the clamped last tile: nothing may land after exit)

  // ---- epilogue: normalise, split into the bf16 operand planes of the consumer GEMM.  Lane: query row q, dv columns
  // 32 dvb + 8 g + 4 half + (0..3) for register group g
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l_tot = l_run[qb];
    {
      const unsigned u = __float_as_uint(l_tot);
      const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      l_tot = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const float inv = 1.f / l_tot;
    const size_t orow = (size_t)b * a.Lq + q0i + wave * 64 + qb * 32 + l31;
#pragma unroll
    for (int dvb = 0; dvb < 2; ++dvb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma clang fp contract(off)
        const float v0 = o[dvb][qb][4 * g] * inv, v1 = o[dvb][qb][4 * g + 1] * inv, v2 = o[dvb][qb][4 * g + 2] * inv, v3 = o[dvb][qb][4 * g + 3] * inv;
        uint2 h, l;
        h.x = cvt_pk_bf16(v0, v1);
        h.y = cvt_pk_bf16(v2, v3);
        l.x = cvt_pk_bf16(v0 - bf_lo(h.x), v1 - bf_hi(h.x));
        l.y = cvt_pk_bf16(v2 - bf_lo(h.y), v3 - bf_hi(h.y));
        const size_t col = (size_t)(dvb * 32 + 8 * g + 4 * half);
        *reinterpret_cast<uint2*>(a.o_hi + orow * a.ldop + col) = h;
        *reinterpret_cast<uint2*>(a.o_lo + orow * a.ldop + col) = l;
      }
  }
}

}  // namespace

int launch_vt_pack32(const float* v, int ldv, int batch, int L, void* vt, hipStream_t st) {
  const size_t n = (size_t)batch * ((L + 31) / 32) * DV;
  hipLaunchKernelGGL(k_vt_pack32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, ldv, batch, L,
                     reinterpret_cast<unsigned short*>(vt));
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

// (default on; DS2_ATTN_X4=0 keeps the 8-wave kernel for A/B runs)
bool attention_x4_enabled() {
  static const bool on = [] { const char* e = getenv("DS2_ATTN_X4"); return !(e && atoi(e) == 0); }();
  return on && DS2_ATTN_K_F16 && ds2_precision() == DS2_PREC_BF16X3K;
}
// shapes the kernel takes: full 256-query blocks, at least half of the chip's CUs busy (fewer: the 8-wave kernel's 128-query form)
bool attention_x4_supported(int batch, int Lq, int Lk, int dv, bool planes_out) {
  return dv == DV && planes_out && Lq % 256 == 0 && Lk >= 3 * BK && Lk % BK == 0 && batch * (Lq / 256) > 128;
}

int launch_attention_x4(const float* q, int ldq, const void* k_f16, const void* vt32, int batch, int Lq, int Lk, float scale,
                        hipStream_t st, void* o_hi, void* o_lo, int ldop, const float* q_rope_cis, int q_rope_grid, bool q_shared) {
  DS2_REQUIRE(attention_x4_supported(batch, Lq, Lk, DV, o_hi && o_lo), "attention_x4: unsupported shape");
  DS2_REQUIRE(ldq % 4 == 0 && ldop % 4 == 0 && q && k_f16 && vt32, "attention_x4: bad argument");
  X4Args a{q, ldq, reinterpret_cast<const char*>(k_f16), reinterpret_cast<const char*>(vt32),
           reinterpret_cast<unsigned short*>(o_hi), reinterpret_cast<unsigned short*>(o_lo), ldop, batch, Lq, Lk, scale,
           q_rope_cis, q_rope_grid, 0, q_shared ? 0 : Lq};
  for (int w = 1; w * w <= q_rope_grid; ++w)
    if (w * w == q_rope_grid) a.rope_w = w;
  DS2_REQUIRE(!q_rope_cis || q_rope_grid > 0, "attention_x4: rope grid");
  if (q_rope_cis) hipLaunchKernelGGL(k_attention_x4<true>, dim3(batch * (Lq / 256)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_attention_x4<false>, dim3(batch * (Lq / 256)), dim3(256), 0, st, a);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
