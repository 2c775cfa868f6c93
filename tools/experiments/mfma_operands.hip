// Does the operand pattern of the attention's score MFMAs cost cycles?  Two chains; B fragments rotate through NB registers
// quadruples held in AGPRs or VGPRs; long and short runs (clock under sustained MFMA load).
#include <stdio.h>
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f16x8 a[3], b[16];
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 8; ++i) a[j][i] = (_Float16)(seed * (threadIdx.x * 0.37f + i + j));
  for (int j = 0; j < 16; ++j) for (int i = 0; i < 8; ++i) b[j][i] = (_Float16)(seed * (0.5f - i * 0.11f + j * threadIdx.x * 0.01f));
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (V == 0) {   // B in AGPRs, 16 different fragments
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a[r % 3]), "a"(b[r]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a[r % 3]), "a"(b[(r + 8) % 16]));
      } else if (V == 1) {   // B in VGPRs
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a[r % 3]), "v"(b[r]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a[r % 3]), "v"(b[(r + 8) % 16]));
      } else {               // one fragment
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a[0]), "v"(b[0]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(const char* what, int iters, float seed) {
  float* out; (void)hipMalloc((void**)&out, 256 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, iters, seed);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, iters, seed);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s iters %5d (%.2f ms/launch) data x%-4g: %.1f cycles per MFMA\n", what, iters, ms / reps, seed, ms * 1e-3 / reps / ((double)iters * 32) * 2.4e9);
  (void)hipFree(out);
}
int main() {
  run<2>("one fragment", 500, 0.f); run<2>("one fragment", 500, 1.f); run<2>("one fragment", 4000, 1.f);
  run<1>("16 B fragments in VGPRs", 500, 0.f); run<1>("16 B fragments in VGPRs", 500, 1.f); run<1>("16 B fragments in VGPRs", 4000, 1.f);
  run<0>("16 B fragments in AGPRs", 500, 0.f); run<0>("16 B fragments in AGPRs", 500, 1.f); run<0>("16 B fragments in AGPRs", 4000, 1.f);
  return 0;
}
