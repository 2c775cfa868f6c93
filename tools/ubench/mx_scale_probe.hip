// Probe of v_mfma_scale_f32_32x32x64_f8f6f4's scale operands: all-ones operands, every lane's scale byte 127 (= 1.0) except ONE lane's,
// and the list of C entries that moved tells which (row, K block) that lane's byte scales.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int OA, int OB>
__global__ void k_probe(const v8i* a, const v8i* b, const int* sa, const int* sb, float* c) {
  const int l = threadIdx.x;
  v16f acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, OA, sa[l], OB, sb[l]);
  for (int r = 0; r < 16; ++r) c[l * 16 + r] = acc[r];
}

int main() {
  v8i *da, *db; int *dsa, *dsb; float* dc;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 16 * 4));
  std::vector<unsigned char> fa(64 * 32, 0x38), fb(64 * 32, 0x38);   // e4m3 1.0
  // make A's K halves distinguishable: lanes 32..63 (second 32-byte group) hold 2.0 (0x40)
  for (int l = 32; l < 64; ++l) for (int j = 0; j < 32; ++j) fa[l * 32 + j] = 0x40;
  CK(hipMemcpy(da, fa.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, fb.data(), 64 * 32, hipMemcpyHostToDevice));
  std::vector<float> c(64 * 16), base(64 * 16);
  for (int which = 0; which < 2; ++which)            // 0: perturb a lane of scale A, 1: of scale B
    for (int byte = 0; byte < 4; ++byte)
      for (int L : {-1, 0, 5, 31, 32, 37, 63}) {
        std::vector<int> sa(64, 0x7f7f7f7f), sb(64, 0x7f7f7f7f);
        if (L >= 0) (which ? sb : sa)[L] = 0x7f7f7f7f + (3 << (8 * byte));      // that byte = 130: x 8
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((k_probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dc, 64 * 16 * 4, hipMemcpyDeviceToHost));
        if (L < 0) { base = c; if (which == 0 && byte == 0) printf("base C[0][0] = %g (expect 32*1 + 32*2 = 96)\n", c[0]); continue; }
        int rows[32] = {}, cols[32] = {}, n = 0; float ratio = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r)
          if (c[l * 16 + r] != base[l * 16 + r]) { rows[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]++; cols[l & 31]++; ++n; ratio = c[l * 16 + r] / base[l * 16 + r]; }
        printf("opsel 0: scale %c lane %2d byte %d -> %4d entries moved (last ratio %.4f); rows:", which ? 'B' : 'A', L, byte, n, ratio);
        for (int i = 0; i < 32; ++i) if (rows[i]) printf(" %d", i);
        printf(" cols:");
        for (int i = 0; i < 32; ++i) if (cols[i]) printf(" %d", i);
        printf("\n");
      }
  return 0;
}
