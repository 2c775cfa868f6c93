// Calibration: sustained v_mfma_f32_32x32x16_f16 rate of the whole chip with NCHAIN independent accumulator chains per wave and
// WAVES waves per SIMD (no memory traffic) - what the clock under MFMA load allows, and whether two chains keep the pipe full.
#include <stdio.h>
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NCHAIN>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f32x16 c[NCHAIN];
  for (int n = 0; n < NCHAIN; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int n = 0; n < NCHAIN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NCHAIN; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCHAIN>
void run(int threads, int iters) {
  float* out; (void)hipMalloc((void**)&out, 256 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<NCHAIN>, dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<NCHAIN>, dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)iters * 8 * NCHAIN * (threads / 64) * 256, t = ms * 1e-3 / reps;
  printf("chains %d waves/SIMD %d: %.1f us, %.2f PFLOP/s, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", NCHAIN, threads / 256, t * 1e6,
         n_mfma * 32768 / t * 1e-15, t / ((double)iters * 8 * NCHAIN * (threads / 256)) * 1e9, t / ((double)iters * 8 * NCHAIN * (threads / 256)) * 2.4e9);
  (void)hipFree(out);
}
int main() {
  run<1>(256, 4000); run<2>(256, 2000); run<4>(256, 1000); run<2>(512, 1000); run<4>(512, 500);
  return 0;
}
