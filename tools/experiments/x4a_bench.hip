// Stand-alone timing of k_attention_x4a at the measured shape (16 objects x 4096 queries x 28 736 keys), for schedule
// experiments: build one binary per generated body (tools/gen/gen_attention_x4a.py OUT --flags) and run them in one gpurun call.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX4A_BODY_INC='"/path/body.inc"' -I det-sam2_amd/csrc -I include x4a_bench.hip -o x4a_bench
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <vector>
#define X4A_BENCH 1
#include "../../det-sam2_amd/csrc/attention_x4a.hip"

void ds2_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
thread_local int t_ds2_precision = -1;
int g_ds2_default_precision = DS2_PREC_BF16X3K;
int launch_w8_merge64(const float*, const float*, int, size_t, void*, void*, int, hipStream_t) { return 0; }

__global__ void k_fill_f16(unsigned short* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed;
  x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
  const float f = ((x & 0xffff) / 65536.0f - 0.5f) * 0.5f;
  p[i] = __builtin_bit_cast(unsigned short, (_Float16)f);
}

int main(int argc, char** argv) {
  const int B = argc > 2 ? atoi(argv[2]) : 16, Lq = 4096, Lk = argc > 1 ? atoi(argv[1]) : 28736, reps = 50;
  const int nkt = (Lk + 31) / 32;
  const size_t kbytes = (size_t)B * Lk * 512 + 32 * 512, vbytes = (size_t)B * nkt * 4096, rows = (size_t)B * Lq;
  char *k, *vt, *qf; float *po, *pml;
  hipMalloc((void**)&k, kbytes); hipMalloc((void**)&vt, vbytes); hipMalloc((void**)&qf, rows / 64 * 32768);
  hipMalloc((void**)&po, rows * 64 * 4); hipMalloc((void**)&pml, rows * 8);
  k_fill_f16<<<(kbytes / 2 + 255) / 256, 256>>>((unsigned short*)k, kbytes / 2, 1);
  k_fill_f16<<<(vbytes / 2 + 255) / 256, 256>>>((unsigned short*)vt, vbytes / 2, 2);
  k_fill_f16<<<(rows / 64 * 16384 + 255) / 256, 256>>>((unsigned short*)qf, rows / 64 * 16384, 3);
  X4AArgs a{k, vt, qf, po, pml, B, Lq, Lk, 1};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(k_attention_x4a, dim3(B * (Lq / 256)), dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_attention_x4a, dim3(B * (Lq / 256)), dim3(256), 0, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h(8);
  hipMemcpy(h.data(), po, 32, hipMemcpyDeviceToHost);
  const double us = ms * 1000.0 / reps, cyc = us * 2400.0 / nkt;
  printf("%s: %.1f us per launch, %.0f cycles(2.4 GHz) per key tile; o[0..3] %g %g %g %g  err=%s\n", argv[0], us, cyc, h[0], h[1], h[2], h[3],
         hipGetErrorString(hipGetLastError()));
  return 0;
}
