// Probe: (1) does MODE.FP16_OVFL (bit 23 of the MODE register) make v_cvt_pk_fp8_f32 / v_cvt_pk_f16_f32 SATURATE instead of producing NaN / inf?
// (2) semantics of v_cvt_scalef32_pk_fp8_f32 (gfx950): is the result cvt(src * scale) or cvt(src / scale)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k(const float* in, unsigned* out, int n, float scale) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const float x = in[i];
  unsigned a, b, c, d, e;
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %1" : "=v"(a) : "v"(x));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(b) : "v"(x));
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 4\n\tv_cvt_pk_fp8_f32 %0, %2, %2\n\tv_cvt_pk_f16_f32 %1, %2, %2\n\ts_nop 1\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0"
               : "=&v"(c), "=&v"(d) : "v"(x));
  asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %1, %2" : "=v"(e) : "v"(x), "v"(scale));
  out[i * 5 + 0] = a; out[i * 5 + 1] = b; out[i * 5 + 2] = c; out[i * 5 + 3] = d; out[i * 5 + 4] = e;
}

int main() {
  const float vals[] = {1.0f, 0.3f, 3.0f, 447.0f, 449.0f, 1000.0f, -1000.0f, 65504.f, 70000.f, -1e6f, 1e-4f, 6.0f};
  const int n = sizeof(vals) / 4;
  float* din; unsigned* dout;
  CK(hipMalloc(&din, 4 * n)); CK(hipMalloc(&dout, 20 * n));
  CK(hipMemcpy(din, vals, 4 * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, n, 4.0f);
  CK(hipDeviceSynchronize());
  unsigned o[5 * 16];
  CK(hipMemcpy(o, dout, 20 * n, hipMemcpyDeviceToHost));
  printf("%12s  fp8(default)  f16(default)  fp8(OVFL=1)  f16(OVFL=1)  scalef32_pk_fp8(x, scale=4)\n", "x");
  for (int i = 0; i < n; ++i)
    printf("%12g  0x%02x          0x%04x        0x%02x         0x%04x       0x%02x\n", vals[i], o[i * 5] & 255, o[i * 5 + 1] & 65535, o[i * 5 + 2] & 255, o[i * 5 + 3] & 65535, o[i * 5 + 4] & 255);
  printf("(e4m3: 1.0 = 0x38, 0.25 = 0x28, 4.0 = 0x48, 448 = 0x7e, NaN = 0x7f; fp16: 65504 = 0x7bff, inf = 0x7c00)\n");
  return 0;
}
