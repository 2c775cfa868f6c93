// Split-precision ("bf16x3") GEMM on the CDNA4 bf16 matrix cores:  C[M,N] = epi(A[M,K] * W[N,K]^T)
//
// fp32 operands are split ONCE, while they are staged into LDS, into two bf16 planes
//     x = x0 + x1 (+ r),  x0 = bf16_rne(x), x1 = bf16_rne(x - x0),  |r| <= 2^-16 |x|
// and the product is accumulated in fp32 as  a0*w0 + a0*w1 + a1*w0  (three v_mfma_f32_32x32x16_bf16 per
// fragment; the dropped a1*w1 and residual terms are O(2^-16)).  That is 3/16 of the fp32-MFMA issue time
// at ~2^-16 relative error per product - measured end-to-end mask error 1-IoU ~ 6e-5 against the fp32
// reference (bar 1e-3), where plain bf16 (the reference's own GPU autocast mode) gives ~5e-3.
//
// Tiling (wave64): 128x128 block tile, BK = 32 fp32 = two 16-deep MFMA k-steps, 4 waves 2x2, each wave
// 64x64 = 2x2 fragments of 32x32 (64 accumulators).  LDS planes are [128 rows][32 k] bf16 with an 80-byte
// row stride: each lane's MFMA operand is one 16-byte ds_read_b128 (8 consecutive k) and the 16-lane read
// groups hit 16 distinct 4-bank slots (conflict-free).  Global->register prefetch of tile t+1 and its
// split/convert VALU work overlap the MFMAs of tile t; two LDS buffers, one barrier per K tile.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                 // bytes per LDS row (64 data + 16 pad)
constexpr int PLANE = BM * ROWB;         // bytes per plane

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// split 4 floats into packed hi (2 dwords) and lo (2 dwords) bf16 planes
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  hi.x = cvt_pk_bf16(v.x, v.y);
  hi.y = cvt_pk_bf16(v.z, v.w);
  const float hx = __uint_as_float(hi.x << 16), hy = __uint_as_float(hi.x & 0xffff0000u);
  const float hz = __uint_as_float(hi.y << 16), hw = __uint_as_float(hi.y & 0xffff0000u);
  lo.x = cvt_pk_bf16(v.x - hx, v.y - hy);
  lo.y = cvt_pk_bf16(v.z - hz, v.w - hw);
}

__global__ __launch_bounds__(256) void k_gemm_nt_bf16x3(GemmArgs g, int mt, int nt) {
  // [buffer][operand A/B][plane hi/lo]
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][2][PLANE];

  const int nwg = mt * nt;
  const int orig = blockIdx.x;
  const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  const int tile_m = wg / nt, tile_n = wg % nt;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[4], rb[4];
  const int nk = (g.K + BK - 1) / BK;

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 3, kq = idx & 7;
      const int k = kt * BK + kq * 4;
      const int m = m0 + row, n = n0 + row;
      ra[i] = (m < g.M && k < g.K) ? *reinterpret_cast<const float4*>(g.A + (size_t)m * g.lda + k)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[i] = (n < g.N && k < g.K) ? *reinterpret_cast<const float4*>(g.W + (size_t)n * g.ldw + k)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 3, kq = idx & 7;
      const int off = row * ROWB + kq * 8;
      uint2 hi, lo;
      split4(ra[i], hi, lo);
      *reinterpret_cast<uint2*>(&lds[buf][0][0][off]) = hi;
      *reinterpret_cast<uint2*>(&lds[buf][0][1][off]) = lo;
      split4(rb[i], hi, lo);
      *reinterpret_cast<uint2*>(&lds[buf][1][0][off]) = hi;
      *reinterpret_cast<uint2*>(&lds[buf][1][1][off]) = lo;
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int koff = s * 32 + half * 16;   // bytes: 16 bf16 per k-step, 8 per half-wave
      bf16x8 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ar = (wm * 64 + t * 32 + l31) * ROWB + koff;
        const int br = (wn * 64 + t * 32 + l31) * ROWB + koff;
        a0[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][0][0][ar]);
        a1[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][0][1][ar]);
        b0[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][1][0][br]);
        b1[t] = *reinterpret_cast<const bf16x8*>(&lds[cur][1][1][br]);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[tm], b0[tn], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[tm], b1[tn], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[tm], b0[tn], acc[tm][tn], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = n0 + wn * 64 + tn * 32 + l31;
    if (n >= g.N) continue;
    const float bias = g.bias ? g.bias[n] : 0.f;
    const float gam = g.gamma ? g.gamma[n] : 1.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + tm * 32 + mfma32_row(e, half);
        if (m >= g.M) continue;
        float v = ds2_act(acc[tm][tn][e] + bias, g.act) * gam;
        if (g.R) {
          const int rm = g.r_mod > 0 ? (m % g.r_mod) : m;
          v += g.R[(size_t)rm * g.ldr + n];
        }
        g.C[(size_t)m * g.ldc + n] = v;
      }
    }
  }
}

}  // namespace

int launch_gemm_bf16x3(const GemmArgs& g, hipStream_t st) {
  DS2_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm: bad dims M=%d N=%d K=%d", g.M, g.N, g.K);
  DS2_REQUIRE(g.K % 4 == 0 && g.lda % 4 == 0 && g.ldw % 4 == 0, "gemm: K/lda/ldw must be multiples of 4 (K=%d lda=%d ldw=%d)",
              g.K, g.lda, g.ldw);
  DS2_REQUIRE((((uintptr_t)g.A) & 15) == 0 && (((uintptr_t)g.W) & 15) == 0, "gemm: A/W must be 16-byte aligned");
  const int mt = cdiv(g.M, BM), nt = cdiv(g.N, BN);
  hipLaunchKernelGGL(k_gemm_nt_bf16x3, dim3(mt * nt), dim3(256), 0, st, g, mt, nt);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
