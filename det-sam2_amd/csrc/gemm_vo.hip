// Output side of the memory cross-attention in ONE kernel (mode bf16x3k with the assembly attention; with or without its key split):
//     o = merge of the attention's un-normalised part(s) (k_w8_merge<64>: weights exp2(m_s - m), one division)  ->  bf16 planes  ->
//     x += o (Wo Wv)^T + (Wo bv + bo)   (the folded value / output projection, 64 -> 256, bf16x3: k_gemm_split_k64)
// instead of k_w8_merge<64> (planes out) -> k_gemm_split_k64<8>: the 64-wide planes never reach HBM, one launch per layer less.
// (RoPEAttention.forward's out_proj on the restructured product of DESIGN.md section 4; memory_attention.py:83-99.)
//
// Same arithmetic, same order per element as the two kernels (the layer outputs were compared bit for bit with the
// two-kernel chain in round 5, profiles/HISTORY.md): v = O * (1 / l) and its bf16 split without contraction, the tile kernels' term order per 16-deep k-step
// (a_lo w_hi, a_hi w_lo, a_hi w_hi), then (acc + bias) + residual.
// Weight-stationary like the K = 64 kernel: the whole weight (256 x 64, two planes = 64 KiB) is staged once per workgroup; a wave owns 32
// token rows, its O fragments come straight from HBM in the MFMA's operand shape; the product is transposed (accumulator lane = token), the
// residual rows are requested before the product and leave as 16-byte pieces (in normal orientation - 4-byte accesses, 256 of them per
// wave and row block - the kernel ran at half the speed of the two it replaces).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int VN = 256, VK = 64, VBR = 128;
constexpr int VPL = 32768;      // lo plane of the weight image (hi at +0): 256 rows of 128 bytes each

__device__ __forceinline__ unsigned v_cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float v_bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float v_bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

struct VoArgs {
  const float* part_o; const float* part_ml;    // [nsplit][rows, 64] un-normalised, [nsplit][rows, 2] (maximum, sum)
  int nsplit;                                   // parts of the key split (few objects: k_attention_x4a's gridDim.y), 1 = none
  const unsigned short *W_hi, *W_lo; int ldw;   // folded projection planes [256, ldw] bf16 (ldw = 64)
  const float* bias;                            // [256]
  const float* R; int ldr, r_mod;               // residual rows (row index modulo r_mod when > 0)
  float* out; int ldo; int rows;
};

__global__ __launch_bounds__(256, 1) void k_vo_merge(VoArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // the weight image: row n at n * 128 B, 16-byte chunk c at (c ^ ((n >> 1) & 7)); 32 pieces of 8 rows per plane, 8 per wave
  {
    const char* wh = reinterpret_cast<const char*>(a.W_hi);
    const char* wl = reinterpret_cast<const char*>(a.W_lo);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int R = (wave * 8 + j) * 8 + (lane >> 3);
      const unsigned off = ((unsigned)R * (unsigned)a.ldw + (unsigned)(((lane & 7) ^ ((R >> 1) & 7)) * 8)) * 2u;
      __builtin_amdgcn_global_load_lds(wh + off, (lds_ptr)(lds + (wave * 8 + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(wl + off, (lds_ptr)(lds + VPL + (wave * 8 + j) * 1024), 16, 0, 0);
    }
  }
  float* bs = reinterpret_cast<float*>(lds + 2 * VPL);   // [256] bias
  bs[tid] = a.bias ? a.bias[tid] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int sw = (l31 >> 1) & 7;

  const int nrb = (a.rows + VBR - 1) / VBR;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int row0 = rb * VBR + wave * 32;
    const int row = row0 + l31;
    const int rowc = row < a.rows ? row : a.rows - 1;
    // ---- O fragments: A operand (row = lane & 31, k = 16 s + 8 half .. + 7); normalised and split as k_w8_merge does for one part
    bf16x8 fh[4], fl[4];
    {
      // k_w8_merge<64>'s expressions: maximum over the parts, weights exp2(m_s - m), weighted sums of the rows and of the row sums
      float m = -INFINITY;
      for (int p = 0; p < a.nsplit; ++p) m = fmaxf(m, a.part_ml[((size_t)p * a.rows + rowc) * 2]);
      float l = 0.f;
      float4 acc[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[s][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p = 0; p < a.nsplit; ++p) {
        const float2 ml = *reinterpret_cast<const float2*>(a.part_ml + ((size_t)p * a.rows + rowc) * 2);
        const float w = __builtin_amdgcn_exp2f(ml.x - m);
        const float* po = a.part_o + ((size_t)p * a.rows + rowc) * VK + half * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const float4 q = *reinterpret_cast<const float4*>(po + s * 16 + jj * 4);
            acc[s][jj].x += q.x * w; acc[s][jj].y += q.y * w; acc[s][jj].z += q.z * w; acc[s][jj].w += q.w * w;
          }
        l += ml.y * w;
      }
      const float inv = 1.f / l;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        unsigned rh[4], rl[4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          {
#pragma clang fp contract(off)
            const float v0 = acc[s][jj].x * inv, v1 = acc[s][jj].y * inv, v2 = acc[s][jj].z * inv, v3 = acc[s][jj].w * inv;
            rh[2 * jj] = v_cvt_pk_bf16(v0, v1);
            rh[2 * jj + 1] = v_cvt_pk_bf16(v2, v3);
            rl[2 * jj] = v_cvt_pk_bf16(v0 - v_bf_lo(rh[2 * jj]), v1 - v_bf_hi(rh[2 * jj]));
            rl[2 * jj + 1] = v_cvt_pk_bf16(v2 - v_bf_lo(rh[2 * jj + 1]), v3 - v_bf_hi(rh[2 * jj + 1]));
          }
        }
        fh[s] = __builtin_bit_cast(bf16x8, (u32x4{rh[0], rh[1], rh[2], rh[3]}));
        fl[s] = __builtin_bit_cast(bf16x8, (u32x4{rl[0], rl[1], rl[2], rl[3]}));
      }
    }
    // residual rows of this lane's token, requested before the product (independent of it): out^T layout - lane (token, half) holds
    // columns 32 t + 8 g + 4 half + e, e = 0..3: one 16-byte piece per (t, g)
    const int rm = a.r_mod > 0 ? (rowc % a.r_mod) : rowc;
    const float* rp = a.R + (size_t)rm * a.ldr + 4 * half;
    float4 res[VN / 32][4];
#pragma unroll
    for (int t = 0; t < VN / 32; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) res[t][g] = *reinterpret_cast<const float4*>(rp + t * 32 + 8 * g);
    f32x16 acc[VN / 32];
#pragma unroll
    for (int t = 0; t < VN / 32; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int coff = (((s * 2 + half) ^ sw) << 4);
#pragma unroll
      for (int t = 0; t < VN / 32; ++t) {
        const unsigned char* wp = lds + (t * 32 + l31) * 128 + coff;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wp);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wp + VPL);
        // transposed (weights as the A operand, accumulator lane = token): the same products in the same order
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, fl[s], acc[t], 0, 0, 0);   // a_lo w_hi
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, fh[s], acc[t], 0, 0, 0);   // a_hi w_lo
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, fh[s], acc[t], 0, 0, 0);   // a_hi w_hi
      }
    }
    // ---- (acc + bias) + residual, one 16-byte store per (t, g)
    if (row < a.rows) {
      float* op = a.out + (size_t)row * a.ldo + 4 * half;
#pragma unroll
      for (int t = 0; t < VN / 32; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bs + t * 32 + 8 * g + 4 * half);
          float4 v = make_float4(acc[t][4 * g + 0] + b4.x, acc[t][4 * g + 1] + b4.y, acc[t][4 * g + 2] + b4.z, acc[t][4 * g + 3] + b4.w);
          v.x *= 1.f; v.y *= 1.f; v.z *= 1.f; v.w *= 1.f;
          v.x += res[t][g].x; v.y += res[t][g].y; v.z += res[t][g].z; v.w += res[t][g].w;
          *reinterpret_cast<float4*>(op + t * 32 + 8 * g) = v;
        }
    }
  }
}

}  // namespace

bool vo_merge_supported(int rows, int ldw) { return rows > 0 && ldw == 64; }

int launch_vo_merge(const float* part_o, const float* part_ml, int nsplit, int rows, const void* w_hi, const void* w_lo, int ldw, const float* bias,
                    const float* R, int ldr, int r_mod, float* out, int ldo, hipStream_t st) {
  DS2_REQUIRE(part_o && part_ml && nsplit >= 1 && w_hi && w_lo && R && out && vo_merge_supported(rows, ldw), "vo_merge: bad argument");
  VoArgs a{};
  a.part_o = part_o; a.part_ml = part_ml; a.nsplit = nsplit; a.W_hi = reinterpret_cast<const unsigned short*>(w_hi); a.W_lo = reinterpret_cast<const unsigned short*>(w_lo);
  a.ldw = ldw; a.bias = bias; a.R = R; a.ldr = ldr; a.r_mod = r_mod; a.out = out; a.ldo = ldo; a.rows = rows;
  static int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int nrb = (rows + VBR - 1) / VBR;
  const int grid = nrb < 2 * ncu ? nrb : 2 * ncu;      // 64 KiB of LDS: two workgroups per CU
  static bool attr_done = false;
  if (!attr_done) {
    DS2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_vo_merge), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(k_vo_merge, dim3(grid), dim3(256), 65536 + 1024, st, a);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
