// Element-wise comparison of k_dwconv7 and k_dwconv7_t4 on random data (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I det-sam2_amd/csrc -I include -o /tmp/dwconv_cmp tools/ubench/dwconv_cmp.hip && /tmp/dwconv_cmp
#include "../../det-sam2_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#include <cstring>
#include <cstdarg>
int launch_split_rows(const float*, int, int, int, void*, void*, int, hipStream_t) { return 1; }   // (not used here)
void ds2_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int main() {
  const int B = 3, H = 64, C = 256;
  const size_t n = (size_t)B * H * H * C;
  std::vector<float> hin(n), hw(49 * C), hb(C);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
  for (auto& v : hin) v = rnd() * 3.f;
  for (auto& v : hw) v = rnd();
  for (auto& v : hb) v = rnd();
  float *in, *w, *b, *o0, *o1;
  hipMalloc(&in, n * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&b, C * 4); hipMalloc(&o0, n * 4); hipMalloc(&o1, n * 4);
  hipMemcpy(in, hin.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dwconv7, grid1((size_t)B * H * (H / 8) * C), dim3(256), 0, 0, in, w, b, o0, B, H, C);
  hipLaunchKernelGGL(k_dwconv7_t4, grid1((size_t)B * (H / 4) * (H / 8) * C), dim3(256), 0, 0, in, w, b, o1, B, H, C);
  std::vector<float> a(n), c(n);
  hipMemcpy(a.data(), o0, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), o1, n * 4, hipMemcpyDeviceToHost);
  size_t bad = 0, first = n;
  double maxd = 0;
  for (size_t i = 0; i < n; ++i)
    if (memcmp(&a[i], &c[i], 4)) { if (first == n) first = i; ++bad; double d = fabs((double)a[i] - c[i]); if (d > maxd) maxd = d; }
  printf("differing elements %zu of %zu, max |diff| %.3e\n", bad, n, maxd);
  if (first < n) {
    const size_t p = first / C; const int cc = first % C, x = p % H, y = (p / H) % H, bb = p / ((size_t)H * H);
    printf("first: b %d y %d x %d c %d: %.9g vs %.9g\n", (int)bb, y, x, cc, a[first], c[first]);
  }
  return 0;
}
