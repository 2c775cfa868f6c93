// What does each instruction class cost a single wave per SIMD next to its MFMAs?  Two v_mfma_f32_32x32x16_f16 chains; variants
// change the operand register files or put N instructions of one class after every MFMA.
#include <stdio.h>
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[4096];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
  unsigned ladr = (threadIdx.x & 63) * 16;
  float4 r0 = {}, r1 = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (V == 1) {        // B operand from the accumulator file
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "a"(b));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "a"(b));
      } else if (V == 2) { // C / D in the accumulator file
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
      } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        if (V == 3) asm volatile(REP4("s_nop 0\n"));
        if (V == 4) asm volatile(REP4("s_add_u32 s40, s40, 1\n") ::: "s40", "scc");
        if (V == 5) asm volatile(REP4("s_waitcnt lgkmcnt(2)\n"));
        if (V == 6) asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n" : "=v"(r0), "=v"(r1) : "v"(ladr));
        if (V == 7) asm volatile(REP4("v_fma_f32 %0, %0, %0, %0\n") : "+v"(x0));                 // a dependent chain
        if (V == 8) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 9) asm volatile(REP8("s_nop 0\n"));
        if (V == 10) asm volatile(REP8("s_add_u32 s40, s40, 1\n") ::: "s40", "scc");
        if (V == 11) asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n" : "+v"(*(double*)&r0), "+v"(*(double*)&r1));
        if (V == 12) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_pk_f16_f32 %1, %2, %3\n v_cvt_pk_f16_f32 %2, %3, %0\n v_cvt_pk_f16_f32 %3, %0, %1\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 13) asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 14) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 15) asm volatile(REP2("v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %4, %5, %1\n v_dot2_f32_f16 %2, %4, %5, %2\n v_dot2_f32_f16 %3, %4, %5, %3\n")
                                  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(ladr), "v"(0x3c003c00));
        if (V == 16) asm volatile(REP2("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")
                                  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 17) asm volatile(REP2("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 18) asm volatile(REP2("v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_pk_f16_f32 %1, %2, %3\n v_cvt_pk_f16_f32 %2, %3, %0\n v_cvt_pk_f16_f32 %3, %0, %1\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 19) asm volatile(REP2("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 20) asm volatile(REP2("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n") : "+v"(*(double*)&r0), "+v"(*(double*)&r1));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
      }
    }
    if (V == 6) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = x0 + x1 + x2 + x3 + r0.x + r1.y;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(const char* what) {
  const int iters = 1000;
  float* out; (void)hipMalloc((void**)&out, 256 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-60s %.1f cycles per PAIR of MFMAs\n", what, ms * 1e-3 / reps / ((double)iters * 8) * 2.4e9);
  (void)hipFree(out);
}
int main() {
  run<0>("plain (VGPR operands)"); run<1>("B operand in AGPRs"); run<2>("C/D in AGPRs");
  run<3>("4 s_nop 0 between"); run<9>("8 s_nop 0 between"); run<4>("4 SALU between"); run<10>("8 SALU between");
  run<5>("4 s_waitcnt between"); run<6>("2 ds_read_b128 between");
  run<7>("4 dependent v_fma between"); run<8>("4 independent v_fma between"); run<11>("4 v_pk_add_f32 between");
  run<12>("4 v_cvt_pk_f16_f32 between"); run<13>("4 v_max3_f32 between"); run<14>("4 v_exp_f32 between");
  run<16>("8 independent v_fma between"); run<15>("8 v_dot2_f32_f16 between"); run<17>("8 v_exp_f32 between");
  run<18>("8 v_cvt_pk_f16_f32 between"); run<19>("8 v_max3_f32 between"); run<20>("4 v_pk_mul_f32 between");
  return 0;
}
