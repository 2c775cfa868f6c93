// Attention with a handful of queries against thousands of keys (the SAM two-way transformer's token -> image
// cross-attention, reference sam2/modeling/sam/transformer.py:265-289 called from :180-191 and :120-127:
// Lq = 8..10 prompt/output tokens, Lk = 4096 image tokens, 8 heads of 16 channels).
//
// A tile-per-block flash kernel has only batch*heads blocks here and walks the 4096 keys serially (190 us for
// 16 objects).  This path splits the KEYS over blocks instead (256 keys per block, one key per thread), in exact
// fp32 VALU arithmetic - the op is 0.3 GFLOP, the matrix cores have nothing to win:
//   k_fewq_part : per (key chunk, head, batch item): scores, chunk-local softmax statistics (m, l) and the
//                 un-normalised chunk output  o = sum_j 2^(s_j - m) v_j
//   k_fewq_merge: log-sum-exp merge of the chunks.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int FQ_KEYS = 256;   // keys per block
constexpr int FQ_MAXQ = 16;    // max queries

template <int D>
__global__ __launch_bounds__(256) void k_fewq_part(AttnArgs a, int S, float* part) {
  __shared__ float qs[FQ_MAXQ][D];
  __shared__ float ps[FQ_MAXQ][FQ_KEYS];
  __shared__ float vs[FQ_KEYS][D + 1];
  __shared__ float red[FQ_MAXQ][4];

  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Lq = a.Lq;
  const float qscale = a.scale * 1.44269504088896340736f;   // fold log2(e): softmax via exp2

  for (int i = tid; i < Lq * D; i += 256) {
    const int t = i / D, d = i % D;
    qs[t][d] = a.q[((size_t)b * Lq + t) * a.ldq + h * D + d] * qscale;
  }
  const int j = s * FQ_KEYS + tid;
  const bool valid = j < a.Lk;
  float kr[D];
  {
    const float* kp = a.k + ((size_t)b * a.Lk + (valid ? j : 0)) * a.ldk + h * D;
    const float* vp = a.v + ((size_t)b * a.Lk + (valid ? j : 0)) * a.ldv + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const float4 k4 = *reinterpret_cast<const float4*>(kp + d);
      kr[d] = k4.x; kr[d + 1] = k4.y; kr[d + 2] = k4.z; kr[d + 3] = k4.w;
      const float4 v4 = *reinterpret_cast<const float4*>(vp + d);
      vs[tid][d] = valid ? v4.x : 0.f; vs[tid][d + 1] = valid ? v4.y : 0.f;
      vs[tid][d + 2] = valid ? v4.z : 0.f; vs[tid][d + 3] = valid ? v4.w : 0.f;
    }
  }
  __syncthreads();

  float sc[FQ_MAXQ];
#pragma unroll
  for (int t = 0; t < FQ_MAXQ; ++t) {
    float acc = 0.f;
    if (t < Lq) {
#pragma unroll
      for (int d = 0; d < D; ++d) acc = fmaf(qs[t][d], kr[d], acc);
    }
    sc[t] = (valid && t < Lq) ? acc : -INFINITY;
    float m = sc[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[t][wave] = m;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < FQ_MAXQ; ++t) {
    if (t < Lq) {
      const float m = fmaxf(fmaxf(red[t][0], red[t][1]), fmaxf(red[t][2], red[t][3]));
      ps[t][tid] = exp2f(sc[t] - m);   // -inf - m -> 0 for padded keys (every chunk holds >= 1 real key)
    }
  }
  __syncthreads();

  // thread (t, d): chunk output and row sum
  float* pb = part + (((size_t)b * a.heads + h) * S + s) * Lq * (D + 2);
  for (int i = tid; i < Lq * D; i += 256) {
    const int t = i / D, d = i % D;
    float o = 0.f, l = 0.f;
#pragma unroll 8
    for (int jj = 0; jj < FQ_KEYS; ++jj) {
      const float p = ps[t][jj];
      o = fmaf(p, vs[jj][d], o);
      l += p;
    }
    pb[t * (D + 2) + 2 + d] = o;
    if (d == 0) {
      pb[t * (D + 2)] = fmaxf(fmaxf(red[t][0], red[t][1]), fmaxf(red[t][2], red[t][3]));
      pb[t * (D + 2) + 1] = l;
    }
  }
}

template <int D>
__global__ void k_fewq_merge(AttnArgs a, int S, const float* part) {
  const int h = blockIdx.x, b = blockIdx.y;
  const int Lq = a.Lq;
  const float* pb = part + ((size_t)b * a.heads + h) * S * Lq * (D + 2);
  for (int i = threadIdx.x; i < Lq * D; i += blockDim.x) {
    const int t = i / D, d = i % D;
    float M = -INFINITY;
    for (int s = 0; s < S; ++s) M = fmaxf(M, pb[((size_t)s * Lq + t) * (D + 2)]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* e = pb + ((size_t)s * Lq + t) * (D + 2);
      const float w = exp2f(e[0] - M);
      L = fmaf(e[1], w, L);
      O = fmaf(e[2 + d], w, O);
    }
    a.o[((size_t)b * Lq + t) * a.ldo + h * D + d] = O / L;
  }
}

float* g_part = nullptr;
size_t g_part_floats = 0;

template <int D>
int launch_t(const AttnArgs& a, hipStream_t st) {
  const int S = cdiv(a.Lk, FQ_KEYS);
  const size_t need = (size_t)a.batch * a.heads * S * a.Lq * (D + 2);
  if (need > g_part_floats) {
    if (g_part) {
      DS2_CHECK_HIP(hipDeviceSynchronize());
      DS2_CHECK_HIP(hipFree(g_part));
      g_part = nullptr;
      g_part_floats = 0;
    }
    DS2_CHECK_HIP(hipMalloc(&g_part, need * sizeof(float)));
    g_part_floats = need;
  }
  hipLaunchKernelGGL((k_fewq_part<D>), dim3(S, a.heads, a.batch), dim3(256), 0, st, a, S, g_part);
  DS2_CHECK_LAUNCH();
  hipLaunchKernelGGL((k_fewq_merge<D>), dim3(a.heads, a.batch), dim3(256), 0, st, a, S, g_part);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

}  // namespace

bool attention_fewq_supported(const AttnArgs& a) {
  return a.win_q == 0 && a.Lq <= FQ_MAXQ && a.D == a.DV && (a.D == 16 || a.D == 32) && a.Lk >= 4 * FQ_KEYS &&
         a.o_hi == nullptr && a.k_pad == nullptr && a.o != nullptr;
}

int launch_attention_fewq(const AttnArgs& a, hipStream_t st) {
  DS2_REQUIRE(attention_fewq_supported(a), "attention_fewq: unsupported shape");
  return a.D == 16 ? launch_t<16>(a, st) : launch_t<32>(a, st);
}
