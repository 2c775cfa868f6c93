// Micro-benchmark of the L2 -> LDS operand path on MI355X: how many bytes per clock per CU a 512-thread block can
// pull through global_load_lds (LDS-DMA) or global_load_dwordx4, as a function of the SHAPE of one wave-instruction's
// footprint:  A = 16 rows x 64 B (the bf16x3 plane tiles of gemm_split_r3), B = 8 rows x 128 B (hi|lo interleaved
// planes), C = 1 KiB contiguous.  Each CU streams its own L2-resident region (96 KiB) over and over.
//   hipcc --offload-arch=gfx950 -O3 -o glds_path glds_path.hip && ./glds_path
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr;

// MODE 0: glds, 1: global_load_dwordx4 to registers (+ ds_write_b128)
template <int SHAPE, int MODE, int INFLIGHT>
__global__ __launch_bounds__(512, 1) void k(const char* __restrict__ src, size_t region, int row_stride, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[131072];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + (size_t)blockIdx.x * region;
  // lane -> byte offset inside one piece
  unsigned lo;
  if (SHAPE == 0) lo = (lane >> 2) * row_stride + (lane & 3) * 16;          // 16 rows x 64 B
  else if (SHAPE == 1) lo = (lane >> 3) * (2 * row_stride) + (lane & 7) * 16;   // 8 rows x 128 B
  else lo = lane * 16;                                                     // 1 KiB contiguous
  const unsigned piece_rows = SHAPE == 0 ? 16 : (SHAPE == 1 ? 8 : 1);
  const unsigned piece_step = SHAPE == 2 ? 1024 : piece_rows * (SHAPE == 1 ? 2 * row_stride : row_stride);
  unsigned acc = 0;
  uint4 r[INFLIGHT];
  unsigned off = wave * piece_step;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
      unsigned o = off + lo + (SHAPE == 2 ? 0 : (j & 3) * 128);   // walk K inside the rows, then next rows
      o = o % (unsigned)(region - 16);
      o &= ~15u;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds(base + o, (lds_ptr)(lds + ((wave * INFLIGHT + j) * 1024)), 16, 0, 0);
      } else {
        r[j] = *reinterpret_cast<const uint4*>(base + o);
      }
      if ((j & 3) == 3 || SHAPE == 2) off += 8 * piece_step;
    }
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int j = 0; j < INFLIGHT; ++j) *reinterpret_cast<uint4*>(lds + ((wave * INFLIGHT + j) * 1024) + lane * 16) = r[j];
    }
    if (off > region) off -= (unsigned)region;
  }
  __syncthreads();
  acc = lds[tid * 4];
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int SHAPE, int MODE, int INFLIGHT>
void run(const char* name, const char* src, size_t region, int row_stride, unsigned* sink) {
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<SHAPE, MODE, INFLIGHT>), dim3(256), dim3(512), 0, 0, src, region, row_stride, 200, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<SHAPE, MODE, INFLIGHT>), dim3(256), dim3(512), 0, 0, src, region, row_stride, iters, sink);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = 256.0 * 8 * INFLIGHT * 1024.0 * iters;
  printf("%-44s inflight/wave %2d  %7.3f ms  %7.2f TB/s  %6.2f B/clk/CU @2.4GHz\n", name, INFLIGHT, ms, bytes / ms * 1e-9,
         bytes / ms * 1e-9 * 1e12 / 256 / 2.4e9);
}

int main() {
  const size_t region = 96 * 1024;
  char* src; unsigned* sink;
  CK(hipMalloc(&src, 256 * region + 65536));
  CK(hipMemset(src, 1, 256 * region + 65536));
  CK(hipMalloc(&sink, 64));
  const int rs = 2304;   // bytes between rows of one plane (lda = 1152 bf16)
  run<0, 0, 8>("glds 16 rows x 64 B", src, region, rs, sink);
  run<1, 0, 8>("glds 8 rows x 128 B", src, region, rs, sink);
  run<2, 0, 8>("glds 1 KiB contiguous", src, region, rs, sink);
  run<0, 0, 16>("glds 16 rows x 64 B", src, region, rs, sink);
  run<1, 0, 16>("glds 8 rows x 128 B", src, region, rs, sink);
  run<2, 0, 16>("glds 1 KiB contiguous", src, region, rs, sink);
  run<0, 1, 8>("global_load_dwordx4 16 rows x 64 B + ds_write", src, region, rs, sink);
  run<1, 1, 8>("global_load_dwordx4 8 rows x 128 B + ds_write", src, region, rs, sink);
  run<2, 1, 8>("global_load_dwordx4 1 KiB contiguous + ds_write", src, region, rs, sink);
  return 0;
}
