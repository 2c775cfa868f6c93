// How exact is v_mfma_scale_f32_32x32x64_f8f6f4?  (a) bits kept below the largest product inside one instruction, (b) bits kept below a large C.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int FA>
__global__ void k(const v8i* a, const v8i* b, const int* sa, const int* sb, const float* cin, float* c) {
  const int l = threadIdx.x;
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = cin[l * 16 + r];
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 16; ++r) c[l * 16 + r] = acc[r];
}

int main() {
  v8i *da, *db; int *dsa, *dsb; float *dc, *dcin;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 4096)); CK(hipMalloc(&dcin, 4096));
  std::vector<float> c(1024), cin(1024, 0.f);
  int fmt_a = 0;
  auto run = [&](const std::vector<unsigned char>& fa, const std::vector<unsigned char>& fb, const std::vector<int>& sa, const std::vector<int>& sb) {
    CK(hipMemcpy(da, fa.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(db, fb.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcin, cin.data(), 4096, hipMemcpyHostToDevice));
    if (fmt_a == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dcin, dc);
    else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dcin, dc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c.data(), dc, 4096, hipMemcpyDeviceToHost));
  };
  // (a) row 0 x col 0: k = 0 holds 1.0 * 1.0 (K block 0, scale 1); k = 32 holds 1.0 * 1.0 * 2^-d (K block 1 scaled by 2^-d); all else zero.
  //     C[0][0] = 1 + 2^-d exactly in fp32 for d <= 23.
  printf("(a) 1 + 2^-d inside one instruction (two products in different K blocks):\n");
  for (int d = 1; d <= 26; ++d) {
    std::vector<unsigned char> fa(2048, 0), fb(2048, 0);
    fa[0 * 32 + 0] = 0x38; fb[0 * 32 + 0] = 0x38;              // lane 0 byte 0  = k 0
    fa[0 * 32 + 16] = 0x38; fb[0 * 32 + 16] = 0x38;            // lane 0 byte 16 = k 32
    std::vector<int> sa(64, 127), sb(64, 127);
    sa[32] = 127 - d;                                           // row 0, K block 1
    std::fill(cin.begin(), cin.end(), 0.f);
    run(fa, fb, sa, sb);
    printf("  d=%2d: got %.9g  expect %.9g  %s\n", d, c[0], 1.0 + ldexp(1.0, -d), c[0] == (float)(1.0 + ldexp(1.0, -d)) ? "exact" : "LOST");
  }
  // (a2) the same with both products in the SAME K block: 1.0 * 1.0 and 2^-6 * 2^-6 ... (e4m3 normals only reach 2^-12 this way)
  printf("(a2) 1 + x*y with both products in one K block (e4m3 values):\n");
  for (int e = 1; e <= 6; ++e) {
    std::vector<unsigned char> fa(2048, 0), fb(2048, 0);
    fa[0] = 0x38; fb[0] = 0x38;
    fa[1] = (unsigned char)((7 - e) << 3); fb[1] = (unsigned char)((7 - e) << 3);      // 2^-e each
    std::vector<int> sa(64, 127), sb(64, 127);
    run(fa, fb, sa, sb);
    printf("  2^-%d * 2^-%d: got %.9g expect %.9g %s\n", e, e, c[0], 1.0 + ldexp(1.0, -2 * e), c[0] == (float)(1.0 + ldexp(1.0, -2 * e)) ? "exact" : "LOST");
  }
  // (b) C = 1.0, one product 2^-d (through the scale): 1 + 2^-d
  printf("(b) C = 1 plus ONE product of 2^-d:\n");
  for (int d = 1; d <= 26; ++d) {
    std::vector<unsigned char> fa(2048, 0), fb(2048, 0);
    fa[0] = 0x38; fb[0] = 0x38;
    std::vector<int> sa(64, 127), sb(64, 127);
    sa[0] = 127 - d;
    std::fill(cin.begin(), cin.end(), 0.f); cin[0] = 1.0f;
    run(fa, fb, sa, sb);
    printf("  d=%2d: got %.9g  expect %.9g  %s\n", d, c[0], 1.0 + ldexp(1.0, -d), c[0] == (float)(1.0 + ldexp(1.0, -d)) ? "exact" : "LOST");
  }
  // (c) C = 1, 64 products of 2^-d each (sum 64 * 2^-d): what a cross term looks like next to a large accumulator
  printf("(c) C = 1 plus 64 products of 2^-d each:\n");
  for (int d = 8; d <= 28; d += 2) {
    std::vector<unsigned char> fa(2048, 0), fb(2048, 0);
    for (int j = 0; j < 32; ++j) { fa[j] = 0x38; fb[j] = 0x38; fa[32 * 32 + j] = 0x38; fb[32 * 32 + j] = 0x38; }   // lanes 0 and 32: row 0 / col 0, all 64 k
    std::vector<int> sa(64, 127), sb(64, 127);
    sa[0] = 127 - d; sa[32] = 127 - d;
    std::fill(cin.begin(), cin.end(), 0.f); cin[0] = 1.0f;
    run(fa, fb, sa, sb);
    const float expect = (float)(1.0 + 64.0 * ldexp(1.0, -d));
    printf("  d=%2d: got %.9g  expect %.9g  %s\n", d, c[0], expect, c[0] == expect ? "exact" : "LOST");
  }
  // (d) subnormal inputs: a = m * 2^-9 (e4m3) resp. m * 2^-16 (e5m2) times b = 1.0
  printf("(d) subnormal A inputs times 1.0:\n");
  std::fill(cin.begin(), cin.end(), 0.f);
  for (fmt_a = 0; fmt_a < 2; ++fmt_a)
    for (int m = 1; m <= (fmt_a ? 3 : 7); ++m) {
      std::vector<unsigned char> fa(2048, 0), fb(2048, 0);
      fa[0] = (unsigned char)m; fb[0] = 0x38;
      std::vector<int> sa(64, 127), sb(64, 127);
      run(fa, fb, sa, sb);
      const float expect = ldexpf((float)m, fmt_a ? -16 : -9);
      printf("  %s subnormal m=%d: got %.9g expect %.9g %s\n", fmt_a ? "e5m2" : "e4m3", m, c[0], expect, c[0] == expect ? "exact" : "DIFFERENT");
    }
  fmt_a = 0;
  // (e) exhaustive: every e4m3 byte a times b = 1.0 against the OCP decode
  int bad = 0;
  for (int v = 0; v < 256; ++v) {
    if ((v & 0x7f) == 0x7f) continue;
    std::vector<unsigned char> fa(2048, 0), fb(2048, 0);
    fa[0] = (unsigned char)v; fb[0] = 0x38;
    std::vector<int> sa(64, 127), sb(64, 127);
    run(fa, fb, sa, sb);
    const int e = (v >> 3) & 15, mm = v & 7;
    float expect = e == 0 ? ldexpf((float)mm, -9) : ldexpf(1.0f + mm / 8.0f, e - 7);
    if (v & 128) expect = -expect;
    if (c[0] != expect) { if (bad < 8) printf("  e4m3 byte 0x%02x: got %.9g expect %.9g\n", v, c[0], expect); ++bad; }
  }
  printf("(e) e4m3 decode over all 254 finite bytes: %d differ\n", bad);
  return 0;
}
