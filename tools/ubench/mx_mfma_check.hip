// Round 6 micro-test ("MX scale-operand layout verified by a micro-test against an fp32 reference"):
// v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 - operand layout, scale semantics, mixed formats, and its rate against the fp16 MFMA.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/mx_mfma_check tools/ubench/mx_mfma_check.hip && tools/ubench/mx_mfma_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// fmt: 0 = e4m3 (OCP fn), 1 = e5m2
static float fp8_to_float(uint8_t v, int fmt) {
  const int sign = v >> 7;
  float r;
  if (fmt == 0) {
    const int e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) r = NAN;
    else r = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  } else {
    const int e = (v >> 2) & 31, m = v & 3;
    if (e == 31) r = m ? NAN : INFINITY;
    else r = e == 0 ? ldexpf((float)m, -16) : ldexpf(1.0f + m / 4.0f, e - 15);
  }
  return sign ? -r : r;
}

template <int FA, int FB>
__global__ void k_one(const v8i* a, const v8i* b, const int* sa, const int* sb, float* c, int opsel_a, int opsel_b) {
  const int l = threadIdx.x;
  v16f acc = {};
  // opsel must be an immediate: dispatch over the four byte selectors of each scale register
#define CASE(OA, OB) if (opsel_a == OA && opsel_b == OB) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, OA, sa[l], OB, sb[l]);
  CASE(0, 0) CASE(1, 0) CASE(2, 0) CASE(3, 0) CASE(0, 1) CASE(0, 2) CASE(0, 3) CASE(1, 2) CASE(3, 3)
#undef CASE
  for (int r = 0; r < 16; ++r) c[l * 16 + r] = acc[r];
}

// rate: NI iterations of 4 independent accumulators per wave, 4 waves per workgroup, one workgroup per CU x `wgs`
template <int MODE>   // 0: v_mfma_f32_32x32x16_f16, 1: scaled f8f6f4 32x32x64 (e4m3 x e4m3), 2: mix of 4 f16 + 2 scaled per step
__global__ __launch_bounds__(256) void k_rate(float* out, int ni, int seed) {
  v16f acc[4] = {};
  v8i a8, b8;
  v8h ah, bh;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 ^ (threadIdx.x * 0x01010101 * (i + seed) & 0x07070707); b8[i] = 0x30303030 ^ ((threadIdx.x + i) * 0x01010101 & 0x03030303); ah[i] = (_Float16)(0.5f + 0.001f * ((threadIdx.x + i * seed) & 63)); bh[i] = (_Float16)(0.25f + 0.002f * ((threadIdx.x * 3 + i) & 31)); }
  const int sc = 0x7f7f7f7f;
  for (int it = 0; it < ni; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[j], 0, 0, 0, sc, 0, sc);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[j], 0, 0, 0, sc, 0, sc);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc[j], 0, 0, 1, sc, 1, sc);
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) out[0] = s;
}

struct Case { const char* name; int fa, fb, opa, opb; bool per_lane_scale; };

int main() {
  srand(7);
  // A [32 rows][64 k], B [32 cols][64 k] as fp8 bytes; scales per (row, k block of 32)
  std::vector<uint8_t> A(32 * 64), B(32 * 64);
  const Case cases[] = {{"e4m3 x e4m3, scale 1", 0, 0, 0, 0, false}, {"e4m3 x e4m3, per-lane scales, opsel 0/0", 0, 0, 0, 0, true},
                        {"e4m3 x e4m3, per-lane scales, opsel 1/2", 0, 0, 1, 2, true}, {"e4m3 x e4m3, per-lane scales, opsel 3/3", 0, 0, 3, 3, true},
                        {"e5m2 (A) x e4m3 (B), per-lane scales, opsel 2/0", 1, 0, 2, 0, true}, {"e4m3 (A) x e5m2 (B), per-lane scales, opsel 0/1", 0, 1, 0, 1, true}};
  v8i *da, *db; int *dsa, *dsb; float* dc;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 16 * 4));
  int fails = 0;
  for (const Case& cs : cases) {
    for (auto& v : A) { do v = rand() & 255; while (isnan(fp8_to_float(v, cs.fa)) || isinf(fp8_to_float(v, cs.fa)) || fabsf(fp8_to_float(v, cs.fa)) > 16.f); }
    for (auto& v : B) { do v = rand() & 255; while (isnan(fp8_to_float(v, cs.fb)) || isinf(fp8_to_float(v, cs.fb)) || fabsf(fp8_to_float(v, cs.fb)) > 16.f); }
    uint8_t ea[32][2], eb[32][2];
    for (int r = 0; r < 32; ++r) for (int kb = 0; kb < 2; ++kb) { ea[r][kb] = cs.per_lane_scale ? 120 + rand() % 12 : 127; eb[r][kb] = cs.per_lane_scale ? 118 + rand() % 14 : 127; }
    // MEASURED layout (tools/ubench/mx_scale_probe.hip): lane l holds row l % 32; its 32 bytes are TWO 16-byte runs of K,
    // byte j -> k = 32 * (j / 16) + 16 * (l / 32) + j % 16; the selected scale byte of lane l scales row l % 32 over the LOGICAL K block
    // l / 32 (k = 32 * (l / 32) .. + 31), i.e. 16 bytes of this lane and 16 bytes of lane l ^ 32
    std::vector<uint8_t> fa(64 * 32), fb(64 * 32);
    std::vector<int> sa(64), sb(64);
    for (int l = 0; l < 64; ++l) {
      for (int j = 0; j < 32; ++j) {
        const int k = 32 * (j / 16) + 16 * (l / 32) + j % 16;
        fa[l * 32 + j] = A[(l % 32) * 64 + k]; fb[l * 32 + j] = B[(l % 32) * 64 + k];
      }
      // the selected byte holds the scale, the other three bytes garbage that must not matter
      unsigned ga = rand(), gb = rand();
      ga = (ga & ~(255u << (8 * cs.opa))) | ((unsigned)ea[l % 32][l / 32] << (8 * cs.opa));
      gb = (gb & ~(255u << (8 * cs.opb))) | ((unsigned)eb[l % 32][l / 32] << (8 * cs.opb));
      sa[l] = (int)ga; sb[l] = (int)gb;
    }
    CK(hipMemcpy(da, fa.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, fb.data(), 64 * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    if (cs.fa == 0 && cs.fb == 0) hipLaunchKernelGGL((k_one<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, cs.opa, cs.opb);
    if (cs.fa == 1 && cs.fb == 0) hipLaunchKernelGGL((k_one<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, cs.opa, cs.opb);
    if (cs.fa == 0 && cs.fb == 1) hipLaunchKernelGGL((k_one<0, 1>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, cs.opa, cs.opb);
    CK(hipDeviceSynchronize());
    std::vector<float> c(64 * 16);
    CK(hipMemcpy(c.data(), dc, 64 * 16 * 4, hipMemcpyDeviceToHost));
    // C layout of every 32x32 MFMA: col (B row) = lane & 31, row (A row) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    double worst = 0, worst_t = 0; int dbg = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31;
      double ref = 0, reft = 0, mag = 0;
      for (int k = 0; k < 64; ++k) {
        mag += fabs((double)fp8_to_float(A[m * 64 + k], cs.fa) * ldexp(1.0, ea[m][k / 32] - 127) * fp8_to_float(B[n * 64 + k], cs.fb) * ldexp(1.0, eb[n][k / 32] - 127));
        ref += (double)fp8_to_float(A[m * 64 + k], cs.fa) * ldexp(1.0, ea[m][k / 32] - 127) * fp8_to_float(B[n * 64 + k], cs.fb) * ldexp(1.0, eb[n][k / 32] - 127);
        reft += (double)fp8_to_float(A[n * 64 + k], cs.fa) * ldexp(1.0, ea[n][k / 32] - 127) * fp8_to_float(B[m * 64 + k], cs.fb) * ldexp(1.0, eb[m][k / 32] - 127);
      }
      if (fabs(c[l * 16 + r] - ref) / (mag + 1e-30) > 1e-6 && dbg < 6) { printf("    m=%d n=%d got %.9g ref %.9g mag %.6g\n", m, n, c[l * 16 + r], ref, mag); ++dbg; }
      worst = fmax(worst, fabs(c[l * 16 + r] - ref) / (mag + 1e-30));          // relative to the sum of |terms|
      worst_t = fmax(worst_t, fabs(c[l * 16 + r] - reft) / (mag + 1e-30));
    }
    const bool ok = worst < 1e-6;
    printf("%-52s err / sum|terms| %.2e (transposed C reading: %.2e)  %s\n", cs.name, worst, worst_t, ok ? "OK" : "MISMATCH");
    fails += !ok;
  }
  // rates
  float* dout; CK(hipMalloc(&dout, 4));
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ni = 20000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL((k_rate<0>), dim3(ncu), dim3(256), 0, 0, dout, ni, 3);
      if (mode == 1) hipLaunchKernelGGL((k_rate<1>), dim3(ncu), dim3(256), 0, 0, dout, ni, 3);
      if (mode == 2) hipLaunchKernelGGL((k_rate<2>), dim3(ncu), dim3(256), 0, 0, dout, ni, 3);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double k_per_it = mode == 0 ? 4 * 16.0 : mode == 1 ? 4 * 64.0 : 16 * 16.0 + 8 * 64.0;
    const double flops = 2.0 * 32 * 32 * k_per_it * ni * 4.0 * ncu;
    const double cyc_per_mfma = ms * 1e-3 * 2.4e9 / (ni * (mode == 2 ? 24.0 : 4.0));
    printf("rate mode %d (%s): %.3f ms, %.0f TFLOP/s on %d CUs, ~%.1f cycles (at 2.4 GHz) per MFMA and SIMD\n", mode,
           mode == 0 ? "f16 32x32x16" : mode == 1 ? "scaled e4m3 32x32x64" : "4 x f16 32x32x16 + 2 x scaled 32x32x64 (one K = 64 step of the 2-MFMA-equivalent product)", ms, flops / (ms * 1e-3) / 1e12, ncu, cyc_per_mfma);
  }
  printf(fails ? "FAILED (%d)\n" : "ALL OK\n", fails);
  return fails != 0;
}
