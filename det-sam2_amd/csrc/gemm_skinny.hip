// Linear layers with a handful of rows (M <= 128: the token side of the two-way transformer, 8 tokens x B objects -
// transformer.py:139-236 - and the small heads around it): C[M,N] = act(A[M,K] W[N,K]^T + bias) * gamma + R in exact fp32.
//
// These GEMMs are pure latency on the tile kernels (one or two workgroups walk K serially behind an operand-split pre-pass:
// 128x256x256 took 30 us, 128x256x2048 80 us).  Here the work is spread over the chip instead: a workgroup owns 64 output
// columns x 8 rows, its waves split K; a lane owns one column and walks its slice of K through the TRANSPOSED weight
// Wt[K,N] (cached per model, so every load is one coalesced 256-byte row), the 8 activation values of a k step are
// one LDS address for the whole wave (broadcast reads of the workgroup's 8 rows, staged once); the per-wave partial sums meet in LDS and are
// added in wave order.  Every row's result depends only on (N, K) - never on M or on the other rows - so a stream sharded
// over ranks (other batch sizes) stays bit-identical.  No operand planes, no pre-pass, fp32 FMA throughout.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SK_ROWS = 8, SK_COLS = 64;

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_skinny_linear(const float* __restrict__ A, int lda, const float* __restrict__ Wt, int N, int K,
                                                           const float* __restrict__ bias, const float* __restrict__ gamma,
                                                           const float* __restrict__ R, int ldr, int r_mod, float* __restrict__ C,
                                                           int ldc, int M, int act) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // As[SK_ROWS][K] (the workgroup's rows), then red[NW][SK_ROWS][64]
  float* As = smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x * SK_COLS + lane, row0 = blockIdx.y * SK_ROWS;
  const int nc = n < N ? n : N - 1;
  for (int i = threadIdx.x; i < SK_ROWS * (K / 4); i += NW * 64) {     // rows past M repeat the last one (never stored)
    const int r = i / (K / 4), c4 = i - r * (K / 4);
    *reinterpret_cast<float4*>(As + r * K + c4 * 4) =
        *reinterpret_cast<const float4*>(A + (size_t)(row0 + r < M ? row0 + r : M - 1) * lda + c4 * 4);
  }
  __syncthreads();
  // this wave's K slice: multiples of 4 (K % 4 == 0), the same split for every M
  const int kq = (K / 4 + NW - 1) / NW * 4;
  const int kbeg = wave * kq, kend = kbeg + kq < K ? kbeg + kq : K;
  float acc[SK_ROWS];
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r) acc[r] = 0.f;
  const float* wp = Wt + nc;
#pragma unroll 4
  for (int k = kbeg; k < kend; k += 4) {
    const float w0 = wp[(size_t)k * N], w1 = wp[(size_t)(k + 1) * N], w2 = wp[(size_t)(k + 2) * N], w3 = wp[(size_t)(k + 3) * N];
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) {
      const float4 a = *reinterpret_cast<const float4*>(As + r * K + k);     // one address for the whole wave: LDS broadcast
      acc[r] = __builtin_fmaf(a.x, w0, acc[r]);
      acc[r] = __builtin_fmaf(a.y, w1, acc[r]);
      acc[r] = __builtin_fmaf(a.z, w2, acc[r]);
      acc[r] = __builtin_fmaf(a.w, w3, acc[r]);
    }
  }
  float* red = smem + SK_ROWS * K;
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r) red[(wave * SK_ROWS + r) * SK_COLS + lane] = acc[r];
  __syncthreads();
  for (int r = wave; r < SK_ROWS; r += NW) {
    const int row = row0 + r;
    if (row >= M || n >= N) continue;
    float v = red[r * SK_COLS + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[(w * SK_ROWS + r) * SK_COLS + lane];
    if (bias) v += bias[n];
    v = ds2_act(v, act);
    if (gamma) v *= gamma[n];
    if (R) v += R[(size_t)(r_mod > 0 ? row % r_mod : row) * ldr + n];
    C[(size_t)row * ldc + n] = v;
  }
}

__global__ void k_transpose_w(const float* __restrict__ W, int ldw, int N, int K, float* __restrict__ Wt) {
  __shared__ float t[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int nn = n0 + j, kk = k0 + threadIdx.x;
    t[j][threadIdx.x] = (nn < N && kk < K) ? W[(size_t)nn * ldw + kk] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int kk = k0 + j, nn = n0 + threadIdx.x;
    if (kk < K && nn < N) Wt[(size_t)kk * N + nn] = t[threadIdx.x][j];
  }
}

}  // namespace

int launch_transpose_w(const float* W, int ldw, int N, int K, float* Wt, hipStream_t st) {
  hipLaunchKernelGGL(k_transpose_w, dim3((K + 31) / 32, (N + 31) / 32), dim3(32, 8), 0, st, W, ldw, N, K, Wt);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

int launch_skinny_linear(const SkinnyArgs& g, hipStream_t st) {
  DS2_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.K % 4 == 0 && g.lda % 4 == 0 && (reinterpret_cast<uintptr_t>(g.A) & 15) == 0,
              "skinny_linear: K and lda must be multiples of 4, A 16-byte aligned");
  const dim3 grid((g.N + SK_COLS - 1) / SK_COLS, (g.M + SK_ROWS - 1) / SK_ROWS);
  DS2_REQUIRE(g.K <= 4096, "skinny_linear: K <= 4096");
  if (g.K >= 1024) {
    const size_t sh = (size_t)(SK_ROWS * g.K + 16 * SK_ROWS * SK_COLS) * 4;
    static const bool attr = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(k_skinny_linear<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    }();
    DS2_REQUIRE(attr, "skinny_linear: could not raise the dynamic LDS limit");
    hipLaunchKernelGGL((k_skinny_linear<16>), grid, dim3(1024), sh, st, g.A, g.lda, g.Wt, g.N, g.K, g.bias, g.gamma, g.R, g.ldr, g.r_mod,
                       g.C, g.ldc, g.M, g.act);
  } else {
    const size_t sh = (size_t)(SK_ROWS * g.K + 4 * SK_ROWS * SK_COLS) * 4;
    hipLaunchKernelGGL((k_skinny_linear<4>), grid, dim3(256), sh, st, g.A, g.lda, g.Wt, g.N, g.K, g.bias, g.gamma, g.R, g.ldr, g.r_mod,
                       g.C, g.ldc, g.M, g.act);
  }
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
