// A14: connected-component labelling (8-connectivity) + hole filling of mask logits.
//
// Replaces the reference's only native op, sam2/_C `get_connected_componnets` (csrc/connected_components.cu:213-289,
// called from sam2/utils/misc.py:48-61) and `fill_holes_in_mask_scores` (misc.py:365-393, called from
// sam2_video_predictor.py:1343-1346 on the low-res logits [B,1,256,256] of every inferred frame).
//
// Contract (what the callers use): labels int32 > 0 and equal exactly within one 8-connected foreground component,
// 0 on background; counts int32 = area of the pixel's component, 0 on background.  Here a component's label is
// (smallest raster index in it) + 1 - deterministic, independent of thread order.  (The reference's block-based
// union-find numbers a component by the top-left pixel of its first 2x2 block and needs even H, W; any H, W here.)
//
// Algorithm: lock-free union-find over pixels in HBM (parents only ever decrease: atomicMin hooking), four passes:
// init -> hook each foreground pixel to its W / NW / N / NE foreground neighbours (covers every 8-neighbour pair
// once) -> flatten + per-root area by atomicAdd -> gather areas.  Integer/index work, HBM-bound: ~20 B per pixel.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int cc_find(const int* parent, int n) {
  int p = parent[n];
  while (p != n) {
    n = p;
    p = parent[n];
  }
  return n;
}

__device__ __forceinline__ void cc_union(int* parent, int a, int b) {
  for (;;) {
    a = cc_find(parent, a);
    b = cc_find(parent, b);
    if (a == b) return;
    if (a > b) {
      const int t = a; a = b; b = t;
    }
    // hook the larger root under the smaller one; if somebody re-parented b meanwhile, retry from what they wrote
    const int old = atomicMin(parent + b, a);
    if (old == b) return;
    b = old;
  }
}

// FG(i): foreground predicate of pixel i, from a uint8 mask or from logits (background of the mask = logit <= 0)
template <bool FROM_LOGITS>
__device__ __forceinline__ bool cc_fg(const void* src, size_t i) {
  if (FROM_LOGITS) return reinterpret_cast<const float*>(src)[i] <= 0.f;
  return reinterpret_cast<const unsigned char*>(src)[i] != 0;
}

template <bool FROM_LOGITS>
__global__ void k_cc_init(const void* src, int* parent, int* area, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  parent[i] = cc_fg<FROM_LOGITS>(src, i) ? (int)i : -1;
  area[i] = 0;
}

__global__ void k_cc_merge(int* parent, int N, int H, int W) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * H * W;
  if (i >= total) return;
  if (parent[i] < 0) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  if (x > 0 && parent[i - 1] >= 0) cc_union(parent, (int)i, (int)i - 1);
  if (y > 0) {
    const size_t up = i - W;
    if (parent[up] >= 0) cc_union(parent, (int)i, (int)up);
    if (x > 0 && parent[up - 1] >= 0) cc_union(parent, (int)i, (int)up - 1);
    if (x + 1 < W && parent[up + 1] >= 0) cc_union(parent, (int)i, (int)up + 1);
  }
}

// labels[i] = root (global index) + 1 for now; area[root] += 1
__global__ void k_cc_flatten(const int* parent, int* labels, int* area, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (parent[i] < 0) {
    labels[i] = 0;
    return;
  }
  const int root = cc_find(parent, (int)i);
  labels[i] = root + 1;
  atomicAdd(area + root, 1);
}

// counts[i] = area of i's component; labels rebased to the image (root index within the image + 1)
__global__ void k_cc_counts(int* labels, const int* area, int* counts, int HW, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int l = labels[i];
  if (l == 0) {
    counts[i] = 0;
    return;
  }
  counts[i] = area[l - 1];
  labels[i] = l - (int)((i / HW) * HW);
}

// hole = background component (logit <= 0) of area <= max_area -> logit 0.1 (misc.py:380-382)
__global__ void k_fill_holes(float* logits, const int* labels, const int* area, int max_area, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int l = labels[i];
  if (l > 0 && area[l - 1] <= max_area) logits[i] = 0.1f;
}

template <bool FROM_LOGITS>
int cc_run(const void* src, int N, int H, int W, int* parent, int* labels, int* area, hipStream_t st) {
  const size_t total = (size_t)N * H * W;
  const int T = 256;
  const unsigned nb = (unsigned)((total + T - 1) / T);
  hipLaunchKernelGGL((k_cc_init<FROM_LOGITS>), dim3(nb), dim3(T), 0, st, src, parent, area, total);
  DS2_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_cc_merge, dim3(nb), dim3(T), 0, st, parent, N, H, W);
  DS2_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_cc_flatten, dim3(nb), dim3(T), 0, st, (const int*)parent, labels, area, total);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

}  // namespace

extern "C" int ds2_connected_components(const uint8_t* mask, int32_t N, int32_t H, int32_t W, int32_t* labels,
                                        int32_t* counts, int32_t* work, void* stream) {
  DS2_REQUIRE(mask && labels && counts && work, "connected_components: null pointer");
  DS2_REQUIRE(N > 0 && H > 0 && W > 0 && (int64_t)N * H * W < (int64_t)1 << 31, "connected_components: bad sizes %d %d %d", N, H, W);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * H * W;
  int* parent = work;          // [total]
  int* area = work + total;    // [total]
  const int rc = cc_run<false>(mask, N, H, W, parent, labels, area, st);
  if (rc != DS2_OK) return rc;
  const unsigned nb = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(k_cc_counts, dim3(nb), dim3(256), 0, st, labels, (const int*)area, counts, H * W, total);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

extern "C" int ds2_fill_holes(float* logits, int32_t N, int32_t H, int32_t W, int32_t max_area, int32_t* work,
                              void* stream) {
  DS2_REQUIRE(logits && work, "fill_holes: null pointer");
  DS2_REQUIRE(max_area > 0, "fill_holes: max_area must be positive (misc.py:371)");
  DS2_REQUIRE(N > 0 && H > 0 && W > 0 && (int64_t)N * H * W < (int64_t)1 << 31, "fill_holes: bad sizes %d %d %d", N, H, W);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * H * W;
  int* parent = work;              // [total]
  int* area = work + total;        // [total]
  int* labels = work + 2 * total;  // [total]
  const int rc = cc_run<true>(logits, N, H, W, parent, labels, area, st);
  if (rc != DS2_OK) return rc;
  const unsigned nb = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(k_fill_holes, dim3(nb), dim3(256), 0, st, logits, (const int*)labels, (const int*)area, max_area, total);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
