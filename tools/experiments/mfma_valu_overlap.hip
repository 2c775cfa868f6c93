// Does a wave's own VALU work run in the shadow of its MFMAs?  NV independent v_fma_f32 (or NV v_exp_f32) after every
// v_mfma_f32_32x32x16_f16 (two accumulator chains), one or two waves per SIMD.
#include <stdio.h>
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, bool EXP>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.25f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[j]));
      }
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[j]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, bool EXP>
void run(int threads, int iters) {
  float* out; (void)hipMalloc((void**)&out, 256 * 512 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<NV, EXP>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<NV, EXP>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double t = ms * 1e-3 / reps, per = t / ((double)iters * 16 * (threads / 256)) * 2.4e9;
  printf("%d %s per MFMA, %d waves/SIMD: %.1f cycles per MFMA per SIMD\n", NV, EXP ? "v_exp" : "v_fma", threads / 256, per);
  (void)hipFree(out);
}
int main() {
  run<0, false>(256, 1000); run<2, false>(256, 1000); run<4, false>(256, 1000); run<6, false>(256, 1000); run<8, false>(256, 1000);
  run<1, true>(256, 1000); run<2, true>(256, 1000);
  run<4, false>(512, 500); run<6, false>(512, 500); run<8, false>(512, 500); run<2, true>(512, 500);
  return 0;
}
