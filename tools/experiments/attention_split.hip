// Memory-attention fast path: bf16x3 flash attention over PRE-SPLIT operands.
//
// attention_bf16x3.hip splits K and V into bf16 planes while staging every 32-key tile - that work is
// repeated by each of the 32 query blocks that stream the same keys, and its branches keep the tile loop from
// being scheduled as straight-line code.  Here the producers do it once:
//   * k_rope_split   : K rows fp32 -> RoPE (position = row mod 4096, first n_rope rows) -> two bf16 planes
//                      [B*Lk][256]
//   * k_vt_split     : V rows fp32 [B,Lk,64-col chunk] -> two bf16 planes of V^T, blocked per 32-key tile and
//                      key-permuted into the MFMA accumulator's row order: [B][tile][plane][64][32]
// and the attention kernel's tile loop is: 10 unconditional 16-byte global loads, 10 ds_write_b128,
// 40 ds_read_b128, 60 v_mfma_f32_32x32x16_bf16, the online softmax and the in-register split of P.
#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int D = 256, DV = 64, BQ = 128, BKEYS = 32;
constexpr int KROWB = D * 2 + 16, KPLANE = BKEYS * KROWB;   // LDS K plane: 32 rows x 528 B
constexpr int VROWB = 80, VPLANE = DV * VROWB;               // LDS V^T plane: 64 rows x 80 B
constexpr int KS = D / 16, NT = DV / 32;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// by-value arguments on purpose: pointer-passed private arrays get promoted to LDS by the AMDGPU backend
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
__device__ __forceinline__ int vt_pos(int key) {   // see attention_bf16x3.hip
  const int h = (key >> 2) & 1, r = (key & 3) + 4 * (key >> 3);
  return 16 * (r >> 3) + 8 * h + (r & 7);
}

// ---- producer 1: (optional RoPE) + split of 256-wide rows.  One thread per 4 consecutive columns.
__global__ void k_rope_split(const float* x, int ldx, const float* cis, int batch, int L, int n_rope, int grid_tokens,
                             uint2* hi, uint2* lo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)batch * L * 64) return;
  const int c4 = (int)(i & 63);
  const size_t row = i >> 6;
  const int t = (int)(row % L);
  float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c4 * 4);
  if (t < n_rope) {   // two complex pairs (apply_rotary_enc, position_encoding.py:196-220)
    const float4 c = *reinterpret_cast<const float4*>(cis + ((size_t)(t % grid_tokens) * 128 + c4 * 2) * 2);
    v = make_float4(v.x * c.x - v.y * c.y, v.x * c.y + v.y * c.x, v.z * c.z - v.w * c.w, v.z * c.w + v.w * c.z);
  }
  uint2 h, l;
  h.x = cvt_pk_bf16(v.x, v.y);
  h.y = cvt_pk_bf16(v.z, v.w);
  l.x = cvt_pk_bf16(v.x - bf_lo(h.x), v.y - bf_hi(h.x));
  l.y = cvt_pk_bf16(v.z - bf_lo(h.y), v.w - bf_hi(h.y));
  hi[i] = h;
  if (lo) lo[i] = l;      // bf16x3k mode: the keys of the scores carry no lo plane
}

// ---- producer 2: V^T tiles.  vt[b][tile][plane][dv 0..63][pos 0..31] (bf16); keys >= L are zero.
__global__ void k_vt_split(const float* v, int ldv, int batch, int L, unsigned short* vt) {
  const int ntile = (L + 31) / 32;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over batch * ntile * 32 keys * 64 dv
  if (i >= (size_t)batch * ntile * 2048) return;
  const int dv = (int)(i & 63), key = (int)((i >> 6) & 31);
  const size_t bt = i >> 11;
  const int tile = (int)(bt % ntile), b = (int)(bt / ntile);
  const int ki = tile * 32 + key;
  const float x = ki < L ? v[((size_t)b * L + ki) * ldv + dv] : 0.f;
  const unsigned h = cvt_pk_bf16(x, 0.f);
  const unsigned l = cvt_pk_bf16(x - bf_lo(h), 0.f);
  unsigned short* base = vt + bt * 2 * 2048;
  const int off = dv * 32 + vt_pos(key);
  base[off] = (unsigned short)(h & 0xffffu);
  base[2048 + off] = (unsigned short)(l & 0xffffu);
}

// ---- the attention kernel
struct SplitArgs {
  const float* q; int ldq;            // fp32 queries [B*Lq, ldq]
  const uint4* k_hi; const uint4* k_lo;   // bf16 planes [B*Lk][256]  (32 uint4 per row)
  const uint4* vt;                    // [B][ntile][2][64][32] bf16 (256 uint4 per plane)
  float* o; int ldo;                  // fp32 out [B*Lq, ldo] (64 columns written)
  int batch, Lq, Lk;
  float scale;
};

__global__ __launch_bounds__(256) void k_attention_split(SplitArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char Kp[2][2][KPLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Vp[2][2][VPLANE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  // XCD-aware placement: blocks b, b+8, ... share an XCD/L2.  Give every XCD whole objects, so the 32 query
  // blocks that stream the same K/V planes hit the same L2.
  const int nqb = a.Lq / BQ;
  const int nblk = a.batch * nqb;
  int bid = blockIdx.x;
  if (nblk % 8 == 0) bid = (bid % 8) * (nblk / 8) + bid / 8;
  const int b = bid / nqb, q0i = (bid % nqb) * BQ;
  const float sc = a.scale * 1.44269504088896340736f;

  bf16x8 q0[KS], q1[KS];
  {
    float* Qs = reinterpret_cast<float*>(&Kp[0][0][0]);   // [32][D+1] floats (fits in one K buffer)
    for (int w = 0; w < 4; ++w) {
      for (int idx = tid; idx < 32 * (D / 4); idx += 256) {
        const int r = idx / (D / 4), c4 = idx - r * (D / 4);
        const float4 v = *reinterpret_cast<const float4*>(a.q + ((size_t)b * a.Lq + q0i + w * 32 + r) * a.ldq + c4 * 4);
        float* dst = Qs + r * (D + 1) + c4 * 4;
        dst[0] = v.x * sc; dst[1] = v.y * sc; dst[2] = v.z * sc; dst[3] = v.w * sc;
      }
      __syncthreads();
      if (wave == w) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float* qr = Qs + l31 * (D + 1) + ks * 16 + half * 8;
          split8(qr[0], qr[1], qr[2], qr[3], qr[4], qr[5], qr[6], qr[7], q0[ks], q1[ks]);
        }
      }
      __syncthreads();
    }
  }

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int nkt = (a.Lk + BKEYS - 1) / BKEYS;
  // per-thread staging geometry (fixed across tiles): K plane = 32 rows x 32 uint4; thread -> rows r0 + 8*i
  const int kpart = tid & 31, krow0 = tid >> 5;
  const int vrow = tid >> 2, vpart = tid & 3;            // V^T plane = 64 rows x 4 uint4
  const size_t kbase = (size_t)b * a.Lk;
  const uint4* vbase = a.vt + (size_t)b * nkt * 512;

  // staging registers are named scalars (not arrays captured by a lambda: those get promoted to LDS)
  uint4 kh0, kh1, kh2, kh3, kl0, kl1, kl2, kl3, rv0, rv1;
#define DS2_LOAD_K(i, H, L)                                          \
  {                                                                  \
    int key = kt_next * BKEYS + krow0 + 8 * (i);                     \
    key = key < a.Lk ? key : a.Lk - 1; /* tail keys are masked */    \
    const size_t g = (kbase + key) * 32 + kpart;                     \
    H = a.k_hi[g];                                                   \
    L = a.k_lo[g];                                                   \
  }
#define DS2_LOAD_TILE(KT)                                            \
  {                                                                  \
    const int kt_next = (KT);                                        \
    DS2_LOAD_K(0, kh0, kl0) DS2_LOAD_K(1, kh1, kl1) DS2_LOAD_K(2, kh2, kl2) DS2_LOAD_K(3, kh3, kl3) \
    rv0 = vbase[(size_t)kt_next * 512 + tid];                        \
    rv1 = vbase[(size_t)kt_next * 512 + 256 + tid];                  \
  }
#define DS2_STORE_K(i, H, L)                                                        \
  *reinterpret_cast<uint4*>(&Kp[buf_][0][(krow0 + 8 * (i)) * KROWB + kpart * 16]) = H; \
  *reinterpret_cast<uint4*>(&Kp[buf_][1][(krow0 + 8 * (i)) * KROWB + kpart * 16]) = L;
#define DS2_STORE_TILE(BUF)                                                         \
  {                                                                                 \
    const int buf_ = (BUF);                                                         \
    DS2_STORE_K(0, kh0, kl0) DS2_STORE_K(1, kh1, kl1) DS2_STORE_K(2, kh2, kl2) DS2_STORE_K(3, kh3, kl3) \
    *reinterpret_cast<uint4*>(&Vp[buf_][0][vrow * VROWB + vpart * 16]) = rv0;       \
    *reinterpret_cast<uint4*>(&Vp[buf_][1][vrow * VROWB + vpart * 16]) = rv1;       \
  }

  DS2_LOAD_TILE(0)
  DS2_STORE_TILE(0)
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) DS2_LOAD_TILE(kt + 1)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const unsigned char* k0p = &Kp[cur][0][l31 * KROWB + half * 16];
    const unsigned char* k1p = &Kp[cur][1][l31 * KROWB + half * 16];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(k0p + ks * 32);
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(k1p + ks * 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q0[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q1[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0[ks], acc, 0, 0, 0);
    }
    if (kt == nkt - 1) {   // only the last tile can contain keys >= Lk
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (kt * BKEYS + mfma32_row(e, half) >= a.Lk) acc[e] = -INFINITY;
    }
    float tmax = acc[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, acc[e]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      acc[e] = __builtin_amdgcn_exp2f(acc[e] - m_new);
      psum += acc[e];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    bf16x8 pb0[2], pb1[2];
    split8(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], pb0[0], pb1[0]);
    split8(acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15], pb0[1], pb1[1]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
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
    if (kt + 1 < nkt) DS2_STORE_TILE(cur ^ 1)
    __syncthreads();
    cur ^= 1;
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  float* op = a.o + ((size_t)b * a.Lq + q0i + wave * 32 + l31) * a.ldo;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) op[t * 32 + mfma32_row(e, half)] = o[t][e] * inv;
}

}  // namespace

int launch_rope_split(const float* x, int ldx, const float* cis, int batch, int L, int n_rope, int grid_tokens,
                      void* hi, void* lo, hipStream_t st) {
  const size_t n = (size_t)batch * L * 64;
  hipLaunchKernelGGL(k_rope_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ldx, cis, batch, L, n_rope,
                     grid_tokens, reinterpret_cast<uint2*>(hi), reinterpret_cast<uint2*>(lo));
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

int launch_vt_split(const float* v, int ldv, int batch, int L, void* vt, hipStream_t st) {
  const size_t n = (size_t)batch * ((L + 31) / 32) * 2048;
  hipLaunchKernelGGL(k_vt_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, ldv, batch, L,
                     reinterpret_cast<unsigned short*>(vt));
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

// q fp32 [B*Lq, ldq] (256 cols), k planes [B*Lk][256] bf16, vt tiles, o fp32 [B*Lq, ldo] (64 cols).
int launch_attention_split(const float* q, int ldq, const void* k_hi, const void* k_lo, const void* vt, float* o, int ldo,
                           int batch, int Lq, int Lk, float scale, hipStream_t st) {
  DS2_REQUIRE(Lq % BQ == 0 && ldq % 4 == 0 && Lk > 0, "attention_split: Lq must be a multiple of 128");
  SplitArgs a{q, ldq, reinterpret_cast<const uint4*>(k_hi), reinterpret_cast<const uint4*>(k_lo),
              reinterpret_cast<const uint4*>(vt), o, ldo, batch, Lq, Lk, scale};
  hipLaunchKernelGGL(k_attention_split, dim3(batch * (Lq / BQ)), dim3(256), 0, st, a);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
