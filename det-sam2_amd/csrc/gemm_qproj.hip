// Query side of the memory cross-attention in ONE kernel (mode bf16x3k with the assembly attention):
//     LayerNorm (norm2)  ->  q_proj (256 -> 256, bf16x3)  ->  RoPE  ->  * scale * log2(e)  ->  fp16, in the assembly kernel's Q fragment order
// instead of k_layernorm_vec (planes out) -> k_gemm_split_pp256 (fp32 q out) -> k_x4a_qprep: three launches and 67 MB of planes, 67 MB of
// fp32 queries written and read back per layer.  (RoPEAttention.forward, sam2/modeling/sam/transformer.py:312-363; norm2 + the query of
// cross_attn_image: memory_attention.py:74-87.)
//
// Same arithmetic, same order per element as the three kernels (compared bit for bit with the three-kernel chain in round 5, profiles/HISTORY.md; tests/test_hip_stages.py: the query fragments through ds2_op_query_fragments):
//   * LayerNorm statistics as the tree k_layernorm_vec's wave_sum butterfly evaluates (see gemm_mlp256.hip, input LayerNorm), the
//     normalised row split into bf16 hi / lo planes in registers;
//   * the product transposed (accumulators = q^T: lane = token, registers = output dims) with the tile kernels' term order per 16-deep
//     k-step: a_lo w_hi, a_hi w_lo, a_hi w_hi (v_mfma_f32_32x32x16_bf16, weights as the A operand - the products commute);
//   * + bias, then k_x4a_qprep's expressions for the rotation, the scale and the saturating fp16 pack.
// The rows of a 32-row weight block are permuted when they are staged (LDS row rho <- output dim delta(rho)) so that a lane's
// accumulator registers 0..7 / 8..15 ARE two Q fragments (8 consecutive dims each) - no lane exchange.
// Structure: 4 waves x 32 tokens per workgroup; the weight streams through a 4-slot LDS ring in 16 tiles of 16 KiB (64 output dims x
// 64 k, hi plane + lo plane, rows of 128 B XOR-swizzled like the GEMM kernels' images) by LDS-DMA, one barrier per tile.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int QD = 256;            // model width (k and n)
constexpr int QBR = 128;           // token rows per workgroup
constexpr int QSLOT = 16384;       // one ring slot: hi plane at +0, lo plane at +8192; 64 rows of 128 bytes (64 k)
constexpr int QLO = 8192;
constexpr int QNS = 6;             // ring slots: the DMA runs 5 tiles ahead, the fragment reads one tile ahead of the MFMAs
constexpr int QTILES = 16;         // (n-group of 64 dims) x (k tile of 64)

__device__ __forceinline__ unsigned q_cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float q_bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float q_bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned q_cvt_pk_f16(float a, float b) {   // v_cvt_pk_f16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{a, b}), f16x2));
}
// accumulator row rho (0..31 inside a block) -> output dim of the block it holds
__host__ __device__ inline int q_delta(int rho) { return 16 * (rho >> 4) + 8 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 1) + (rho & 3); }

struct QprojArgs {
  const float* x; int ldx; int rows;            // un-normalised fp32 rows [rows, ldx]
  const float *ln_w, *ln_b; float ln_eps;       // norm2
  const unsigned short *W_hi, *W_lo; int ldw;   // q_proj weight planes [256, ldw] bf16
  const float* bias;                            // [256]
  const float* cis; int rope_grid, rope_w;      // axial RoPE table (nullable), tokens per image, image width
  float sc;                                     // scale * log2(e)
  uint4* qfrag; int nrep; size_t rep_stride;    // Q fragments; the result is written nrep times, rep_stride uint4 apart (shared queries)
};

__global__ __launch_bounds__(256, 1) void k_qproj_x4a(QprojArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  float* lns = reinterpret_cast<float*>(lds + QNS * QSLOT);   // [3][256]: ln_w, ln_b, bias
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  lns[tid] = a.ln_w[tid];
  lns[QD + tid] = a.ln_b[tid];
  lns[2 * QD + tid] = a.bias ? a.bias[tid] : 0.f;

  // DMA: a piece = 8 rows x 128 B of one plane; a tile has 8 pieces per plane, wave w issues pieces 2 w, 2 w + 1 of both planes.
  // LDS row R of a tile (0..63) holds output dim 32 (R >> 5) + delta(R & 31) of the tile's n-group.
  const char* wh = reinterpret_cast<const char*>(a.W_hi);
  const char* wl = reinterpret_cast<const char*>(a.W_lo);
  unsigned off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = (wave * 2 + j) * 8 + (lane >> 3);
    const int src_row = 32 * (R >> 5) + q_delta(R & 31);
    off[j] = ((unsigned)src_row * (unsigned)a.ldw + (unsigned)(((lane & 7) ^ ((R >> 1) & 7)) * 8)) * 2u;
  }
  const int sw = (l31 >> 1) & 7;
  // tile T = 4 ng + kt: rows 64 ng .., k 64 kt ..
#define Q_DMA(T)                                                                                                          \
  {                                                                                                                       \
    unsigned char* base_ = lds + ((T) % QNS) * QSLOT;                                                                     \
    const unsigned o_ = (unsigned)((T) >> 2) * 64u * (unsigned)a.ldw * 2u + (unsigned)((T) & 3) * 128u;                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                       \
      __builtin_amdgcn_global_load_lds(wh + (off[j] + o_), (lds_ptr)(base_ + (wave * 2 + j) * 1024), 16, 0, 0);           \
      __builtin_amdgcn_global_load_lds(wl + (off[j] + o_), (lds_ptr)(base_ + QLO + (wave * 2 + j) * 1024), 16, 0, 0);     \
    }                                                                                                                     \
  }
  __syncthreads();   // lns

  const int nrb = (a.rows + QBR - 1) / QBR;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int tok = rb * QBR + wave * 32 + l31;
    const int tokc = tok < a.rows ? tok : a.rows - 1;
    // ---- norm2 of this lane's half row (columns 16 s + 8 half + j) -> bf16 planes as MFMA B fragments (token = lane & 31, k = 16 s + 8 half ..)
    bf16x8 xh[QD / 16], xl[QD / 16];
    {
      const float* px = a.x + (size_t)tokc * a.ldx + half * 8;
      float4 v[QD / 16][2];
#pragma unroll
      for (int s = 0; s < QD / 16; ++s) {
        v[s][0] = *reinterpret_cast<const float4*>(px + s * 16);
        v[s][1] = *reinterpret_cast<const float4*>(px + s * 16 + 4);
      }
      auto both = [](float x) {          // x + (the same quantity of lane ^ 32)
        const unsigned u = __float_as_uint(x);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
      };
      auto tree = [&](float (&p)[QD / 16][2]) {   // wave_sum's xor 32, 16, 8, 4 | 2 (other half-lane) | 1 over the row's 64 float4 groups
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1)
#pragma unroll
          for (int s = 0; s < d; ++s) {
            p[s][0] = p[s][0] + p[s + d][0];
            p[s][1] = p[s][1] + p[s + d][1];
          }
        const float t0 = both(p[0][0]), t1 = both(p[0][1]);
        return t0 + t1;
      };
      float ps[QD / 16][2];
#pragma unroll
      for (int s = 0; s < QD / 16; ++s)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ps[s][jj] = (v[s][jj].x + v[s][jj].y) + (v[s][jj].z + v[s][jj].w);
      const float mean = tree(ps) / (float)QD;
#pragma unroll
      for (int s = 0; s < QD / 16; ++s)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const float d0 = v[s][jj].x - mean, d1 = v[s][jj].y - mean, d2 = v[s][jj].z - mean, d3 = v[s][jj].w - mean;
          ps[s][jj] = 0.f + (__builtin_fmaf(d0, d0, d1 * d1) + __builtin_fmaf(d2, d2, d3 * d3));   // (k_layernorm_vec's contraction, spelled out)
        }
      const float rstd = 1.f / sqrtf(tree(ps) / (float)QD + a.ln_eps);
#pragma unroll
      for (int s = 0; s < QD / 16; ++s) {
        unsigned rh[4], rl[4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int c = 16 * s + 8 * half + 4 * jj;
          const float4 w4 = *reinterpret_cast<const float4*>(lns + c), b4 = *reinterpret_cast<const float4*>(lns + QD + c);
          float4 o;
          o.x = (v[s][jj].x - mean) * rstd * w4.x + b4.x;
          o.y = (v[s][jj].y - mean) * rstd * w4.y + b4.y;
          o.z = (v[s][jj].z - mean) * rstd * w4.z + b4.z;
          o.w = (v[s][jj].w - mean) * rstd * w4.w + b4.w;
          rh[2 * jj] = q_cvt_pk_bf16(o.x, o.y);
          rh[2 * jj + 1] = q_cvt_pk_bf16(o.z, o.w);
          rl[2 * jj] = q_cvt_pk_bf16(o.x - q_bf_lo(rh[2 * jj]), o.y - q_bf_hi(rh[2 * jj]));
          rl[2 * jj + 1] = q_cvt_pk_bf16(o.z - q_bf_lo(rh[2 * jj + 1]), o.w - q_bf_hi(rh[2 * jj + 1]));
        }
        xh[s] = __builtin_bit_cast(bf16x8, (u32x4{rh[0], rh[1], rh[2], rh[3]}));
        xl[s] = __builtin_bit_cast(bf16x8, (u32x4{rl[0], rl[1], rl[2], rl[3]}));
      }
    }
    // ---- ring prologue; fragments of TWO tiles: tile T's MFMAs run on one set while tile T + 1's reads fill the other
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the ring's counted waits assume only DMA in flight)
    Q_DMA(0) Q_DMA(1) Q_DMA(2) Q_DMA(3) Q_DMA(4)
    const int t = tokc % a.rope_grid;
    const size_t blk64 = (size_t)(tok >> 6);
    const int qb = (tok >> 5) & 1;
    bf16x8 fh[2][4][2], fl[2][4][2];
#define Q_READ(T, BUF)                                                                                                    \
  {                                                                                                                       \
    const unsigned char* base_ = lds + ((T) % QNS) * QSLOT;                                                               \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                         \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                                     \
        const unsigned char* r_ = base_ + (b * 32 + l31) * 128 + (((s * 2 + half) ^ sw) << 4);                            \
        fh[BUF][s][b] = *reinterpret_cast<const bf16x8*>(r_);                                                             \
        fl[BUF][s][b] = *reinterpret_cast<const bf16x8*>(r_ + QLO);                                                       \
      }                                                                                                                   \
  }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile 0 (tiles 1..4 may stay in flight)
    __builtin_amdgcn_s_barrier();
    Q_READ(0, 0)
#pragma unroll 1     // (unrolled, hipcc hoists the epilogues' addresses over the groups and spills)
    for (int ng = 0; ng < 4; ++ng) {
      f32x16 acc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int T = ng * 4 + kt;
        // step T: tile T + 1 landed (this wave's pieces: the count; everybody's: the barrier) -> the DMA of tile T + 5 into the slot of
        // tile T - 1 (consumed before anybody reached this barrier) -> tile T + 1's fragment reads -> tile T's MFMAs
        const int rem = QTILES - 2 - T;
        if (rem >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (rem == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (T + 5 < QTILES) Q_DMA(T + 5)
        if (T + 1 < QTILES) Q_READ(T + 1, (kt + 1) & 1)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int ks = kt * 4 + s, cb = kt & 1;
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[cb][s][b], xl[ks], acc[b], 0, 0, 0);   // a_lo w_hi
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[cb][s][b], xh[ks], acc[b], 0, 0, 0);   // a_hi w_lo
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[cb][s][b], xh[ks], acc[b], 0, 0, 0);   // a_hi w_hi
          }
      }
      // ---- the 64 dims of this n-group: registers 0..7 / 8..15 of block b are the fragments ks = 2 (2 ng + b), + 1 of this lane
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int ks = 2 * (2 * ng + b) + f, d0 = 16 * ks + 8 * half;
          const float4 bb0 = *reinterpret_cast<const float4*>(lns + 2 * QD + d0), bb1 = *reinterpret_cast<const float4*>(lns + 2 * QD + d0 + 4);
          float4 v0 = make_float4(acc[b][8 * f + 0] + bb0.x, acc[b][8 * f + 1] + bb0.y, acc[b][8 * f + 2] + bb0.z, acc[b][8 * f + 3] + bb0.w);
          float4 v1 = make_float4(acc[b][8 * f + 4] + bb1.x, acc[b][8 * f + 5] + bb1.y, acc[b][8 * f + 6] + bb1.z, acc[b][8 * f + 7] + bb1.w);
          if (a.cis) {   // apply_rotary_enc on the complex pairs (d, d + 1): k_x4a_qprep's expressions
            const int pair0 = ks * 8 + half * 4;
            int tt = t;
            if (a.rope_w > 0) tt = pair0 < 64 ? t % a.rope_w : t - t % a.rope_w;
            const float4 c0 = *reinterpret_cast<const float4*>(a.cis + ((size_t)tt * 128 + pair0) * 2);
            const float4 c1 = *reinterpret_cast<const float4*>(a.cis + ((size_t)tt * 128 + pair0 + 2) * 2);
            // explicit fused forms: what hipcc's contraction + packing (v_pk_mul_f32 / v_pk_fma_f32 over the two pairs of a float4) makes of
            // k_x4a_qprep's `x cx - y cy`, `x cy + y cx`: the real parts fma(x, cx, -(y cy)); the imaginary part of the FIRST pair
            // fma(x, cy, y cx), of the SECOND pair fma(y, cx, x cy) (tools/qfrag_check.py: every other choice flips ~30 of 2 M fp16 roundings)
            auto re = [](float x, float y, float cx, float cy) { return __builtin_fmaf(x, cx, -(y * cy)); };
            auto im1 = [](float x, float y, float cx, float cy) { return __builtin_fmaf(x, cy, y * cx); };
            auto im2 = [](float x, float y, float cx, float cy) { return __builtin_fmaf(y, cx, x * cy); };
            v0 = make_float4(re(v0.x, v0.y, c0.x, c0.y), im1(v0.x, v0.y, c0.x, c0.y), re(v0.z, v0.w, c0.z, c0.w), im2(v0.z, v0.w, c0.z, c0.w));
            v1 = make_float4(re(v1.x, v1.y, c1.x, c1.y), im1(v1.x, v1.y, c1.x, c1.y), re(v1.z, v1.w, c1.z, c1.w), im2(v1.z, v1.w, c1.z, c1.w));
          }
          auto pk = [](float x, float y) { return q_cvt_pk_f16(ds2_sat_f16(x), ds2_sat_f16(y)); };
          const float sc = a.sc;
          const uint4 out = make_uint4(pk(v0.x * sc, v0.y * sc), pk(v0.z * sc, v0.w * sc), pk(v1.x * sc, v1.y * sc), pk(v1.z * sc, v1.w * sc));
          if (tok < a.rows) {
            uint4* dst = a.qfrag + ((blk64 * 32 + (size_t)(qb * 16 + ks)) * 64 + lane);
            for (int r = 0; r < a.nrep; ++r) dst[(size_t)r * a.rep_stride] = out;
          }
        }
    }
    __builtin_amdgcn_s_barrier();   // the ring restarts: nobody still reads the last tiles
  }
}

}  // namespace

bool qproj_x4a_supported(int rows, int ldx, int ldw) { return rows > 0 && rows % 64 == 0 && ldx % 4 == 0 && ldw % 64 == 0; }

// x [rows, ldx] fp32 -> qfrag (k_x4a_qprep's layout) for rows tokens; nrep > 1: the same rows written for nrep objects (rows = Lq)
int launch_qproj_x4a(const float* x, int ldx, int rows, const float* ln_w, const float* ln_b, float ln_eps, const void* w_hi, const void* w_lo,
                     int ldw, const float* bias, const float* cis, int rope_grid, float scale, void* qfrag, int nrep, hipStream_t st) {
  DS2_REQUIRE(x && ln_w && ln_b && w_hi && w_lo && qfrag && nrep >= 1 && qproj_x4a_supported(rows, ldx, ldw), "qproj_x4a: bad argument");
  int rope_w = 0;
  for (int i = 1; i * i <= rope_grid; ++i)
    if (i * i == rope_grid) rope_w = i;
  DS2_REQUIRE(!cis || rope_grid > 0, "qproj_x4a: rope grid");
  QprojArgs a{};
  a.x = x; a.ldx = ldx; a.rows = rows; a.ln_w = ln_w; a.ln_b = ln_b; a.ln_eps = ln_eps;
  a.W_hi = reinterpret_cast<const unsigned short*>(w_hi); a.W_lo = reinterpret_cast<const unsigned short*>(w_lo); a.ldw = ldw;
  a.bias = bias; a.cis = cis; a.rope_grid = rope_grid > 0 ? rope_grid : 1; a.rope_w = rope_w;
  a.sc = scale * 1.44269504088896340736f;
  a.qfrag = reinterpret_cast<uint4*>(qfrag); a.nrep = nrep; a.rep_stride = (size_t)(rows / 64) * 32 * 64;
  static int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int nrb = (rows + QBR - 1) / QBR;
  const int grid = nrb < ncu ? nrb : ncu;
  const size_t smem = (size_t)QNS * QSLOT + 3 * QD * 4;
  static bool attr_done = false;
  if (!attr_done) {
    DS2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qproj_x4a), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(k_qproj_x4a, dim3(grid), dim3(256), smem, st, a);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
