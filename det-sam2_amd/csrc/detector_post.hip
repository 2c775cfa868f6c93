// F4, detector half: the post-processing between the YOLOv8 head and VideoProcessor.detect_predict
// (det_sam2_RT.py:228-244 reads result.boxes -> xyxy / cls / conf).  In the reference this is ultralytics 8.2.82
// `ops.non_max_suppression` (+ xywh2xyxy, scale_boxes, clip_boxes) over torchvision.ops.nms - CUDA / CPU code of two
// third-party packages.  Here: five small kernels on the caller's stream, bit-exact against oracle/yolo_post.py (which
// restates the published algorithm; parity with ultralytics itself is unpinned offline):
//   k_yolo_score   : per anchor, best class score / first arg-max over the nc score planes (coalesced along anchors)
//   k_yolo_compact : per image, the anchors with conf > conf_thres in ascending anchor order (block scan; deterministic)
//   k_yolo_rank    : rank by counting - conf descending, ties by ascending anchor - and scatter of the sorted records
//                    (xywh -> xyxy, + class * 7680 offset so that classes never suppress each other)
//   k_yolo_mask    : 64 x 64 tiles of the upper-triangular suppression matrix, IoU > iou_thres, one bit per pair
//   k_yolo_sweep   : the greedy pass (one wave per image): visit in order, keep if not yet removed, OR in the row;
//                    writes up to max_det detections mapped back to the original image (letterbox undo + clip)
// Integer / comparison work and a few fp32 operations without contraction: HBM- and latency-bound, nothing for the MFMAs.
#include "common.h"
#include "../../include/detsam2_hip.h"

namespace {

constexpr int YCAP = 8192;            // candidates per image the NMS stage handles (conf > conf_thres); more is an error
constexpr int YWORDS = YCAP / 64;
constexpr float MAX_WH = 7680.f;

struct YoloWork {                     // per-image slices of the caller's work buffer
  float* conf; int* cls;              // [N]
  int* cand; float* cconf;            // [YCAP] anchors over the threshold (ascending), their confidences
  int* count;                         // [1] (+ padding)
  float* sbox; float* soff;           // [YCAP][4] sorted boxes (network pixels) / the same + class offset
  float* sconf; int* scls;            // [YCAP]
  unsigned long long* mask;           // [YCAP][YWORDS]
};
// conf / cls segments rounded up to 16 bytes so that the float4 arrays behind them (sbox, soff) are 16-byte aligned for
// every N (YOLOv8 at 672 x 672 has N = 9261, odd)
__host__ __device__ inline size_t yolo_seg_n(int N) { return (((size_t)N * 4) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t yolo_work_per_image(int N) {
  size_t b = 2 * yolo_seg_n(N) + (size_t)YCAP * 8 + 256 + (size_t)YCAP * 32 + (size_t)YCAP * 8 + (size_t)YCAP * YWORDS * 8;
  return (b + 255) & ~(size_t)255;
}
__device__ __forceinline__ YoloWork yolo_slice(unsigned char* work, int N, int b) {
  unsigned char* p = work + (size_t)b * yolo_work_per_image(N);
  YoloWork w;
  w.conf = reinterpret_cast<float*>(p); p += yolo_seg_n(N);
  w.cls = reinterpret_cast<int*>(p); p += yolo_seg_n(N);
  w.cand = reinterpret_cast<int*>(p); p += (size_t)YCAP * 4;
  w.cconf = reinterpret_cast<float*>(p); p += (size_t)YCAP * 4;
  w.count = reinterpret_cast<int*>(p); p += 256;
  w.sbox = reinterpret_cast<float*>(p); p += (size_t)YCAP * 16;
  w.soff = reinterpret_cast<float*>(p); p += (size_t)YCAP * 16;
  w.sconf = reinterpret_cast<float*>(p); p += (size_t)YCAP * 4;
  w.scls = reinterpret_cast<int*>(p); p += (size_t)YCAP * 4;
  w.mask = reinterpret_cast<unsigned long long*>(p);
  return w;
}

__global__ void k_yolo_score(const float* pred, int nc, int N, unsigned char* work) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (a >= N) return;
  const float* p = pred + ((size_t)b * (4 + nc) + 4) * N + a;
  float best = p[0];
  int arg = 0;
  for (int c = 1; c < nc; ++c) {
    const float s = p[(size_t)c * N];
    if (s > best) { best = s; arg = c; }      // first maximum (torch.max / numpy argmax)
  }
  const YoloWork w = yolo_slice(work, N, b);
  w.conf[a] = best;
  w.cls[a] = arg;
}

__global__ __launch_bounds__(1024) void k_yolo_compact(int N, float conf_thres, unsigned char* work) {
  __shared__ int wsum[16];
  __shared__ int base;
  const YoloWork w = yolo_slice(work, N, blockIdx.x);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int a0 = 0; a0 < N; a0 += 1024) {
    const int a = a0 + tid;
    const float c = a < N ? w.conf[a] : 0.f;
    const int f = (a < N && c > conf_thres) ? 1 : 0;
    const unsigned long long bal = __ballot(f);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int before = base;
    for (int i = 0; i < wave; ++i) before += wsum[i];
    const int pos = before + in_wave;
    if (f && pos < YCAP) { w.cand[pos] = a; w.cconf[pos] = c; }
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int i = 0; i < 16; ++i) t += wsum[i];
      base += t;
    }
    __syncthreads();
  }
  if (tid == 0) *w.count = base;      // may exceed YCAP: reported by the sweep
}

__global__ void k_yolo_rank(const float* pred, int nc, int N, unsigned char* work) {
  const int b = blockIdx.y;
  const YoloWork w = yolo_slice(work, N, b);
  const int n = *w.count <= YCAP ? *w.count : 0;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float ck = w.cconf[k];
  int rank = 0;
  for (int j = 0; j < n; ++j) {        // candidates are in ascending anchor order: j < k <=> anchor_j < anchor_k
    const float cj = w.cconf[j];
    rank += (cj > ck || (cj == ck && j < k)) ? 1 : 0;
  }
  const int a = w.cand[k];
  const float* p = pred + (size_t)b * (4 + nc) * N + a;
  const float x = p[0], y = p[(size_t)N], hw = p[2 * (size_t)N] / 2.f, hh = p[3 * (size_t)N] / 2.f;   // xywh2xyxy
  const float x1 = x - hw, y1 = y - hh, x2 = x + hw, y2 = y + hh;
  const int cls = w.cls[a];
  const float off = (float)cls * MAX_WH;
  reinterpret_cast<float4*>(w.sbox)[rank] = make_float4(x1, y1, x2, y2);
  reinterpret_cast<float4*>(w.soff)[rank] = make_float4(x1 + off, y1 + off, x2 + off, y2 + off);
  w.sconf[rank] = ck;
  w.scls[rank] = cls;
}

// torchvision's nms kernel arithmetic, fp32, no fused multiply-add (the oracle computes the same operations in numpy)
__device__ __forceinline__ bool yolo_iou_gt(const float4 a, const float4 b, float thr) {
#pragma clang fp contract(off)
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = w * h;
  const float sa = (a.z - a.x) * (a.w - a.y), sb = (b.z - b.x) * (b.w - b.y);
  return inter / ((sa + sb) - inter) > thr;
}

__global__ __launch_bounds__(64) void k_yolo_mask(int N, float iou_thres, unsigned char* work) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  const YoloWork w = yolo_slice(work, N, b);
  const int n = *w.count <= YCAP ? *w.count : 0;
  if (rb * 64 >= n || cb * 64 >= n || cb < rb) return;
  __shared__ float4 cbox[64];
  const int t = threadIdx.x;
  const int j0 = cb * 64;
  if (j0 + t < n) cbox[t] = reinterpret_cast<const float4*>(w.soff)[j0 + t];
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  const float4 me = reinterpret_cast<const float4*>(w.soff)[i];
  unsigned long long bits = 0;
  const int jn = n - j0 < 64 ? n - j0 : 64;
  for (int jj = 0; jj < jn; ++jj)
    if (j0 + jj > i && yolo_iou_gt(me, cbox[jj], iou_thres)) bits |= 1ull << jj;
  w.mask[(size_t)i * YWORDS + cb] = bits;
}

__global__ __launch_bounds__(64) void k_yolo_sweep(int N, int max_det, const float* scale, float* dets, int* counts, unsigned char* work) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const YoloWork w = yolo_slice(work, N, b);
  const int total = *w.count;
  float* out = dets + (size_t)b * max_det * 6;
  if (total > YCAP) {                 // more candidates than the NMS stage holds: flag it (the host wrapper raises)
    if (lane == 0) counts[b] = -1;
    return;
  }
  const int n = total, nw = (n + 63) / 64;
  __shared__ unsigned long long removed[YWORDS];
  for (int i = lane; i < YWORDS; i += 64) removed[i] = 0;
  __syncthreads();
  int kept = 0;
  for (int i = 0; i < n && kept < max_det; ++i) {
    const bool gone = (removed[i >> 6] >> (i & 63)) & 1ull;
    __syncthreads();
    if (!gone) {
      for (int c = (i >> 6) + lane; c < nw; c += 64) removed[c] |= w.mask[(size_t)i * YWORDS + c];   // (words left of i's were never written)
      if (lane == 0) {
        float4 bx = reinterpret_cast<const float4*>(w.sbox)[i];
        if (scale) {                  // scale_boxes + clip_boxes: (x - pad) / gain, clipped to the original image
          const float gain = scale[0], px = scale[1], py = scale[2], ow = scale[3], oh = scale[4];
          bx.x = fminf(fmaxf((bx.x - px) / gain, 0.f), ow);
          bx.z = fminf(fmaxf((bx.z - px) / gain, 0.f), ow);
          bx.y = fminf(fmaxf((bx.y - py) / gain, 0.f), oh);
          bx.w = fminf(fmaxf((bx.w - py) / gain, 0.f), oh);
        }
        float* o = out + (size_t)kept * 6;
        o[0] = bx.x; o[1] = bx.y; o[2] = bx.z; o[3] = bx.w; o[4] = w.sconf[i]; o[5] = (float)w.scls[i];
      }
      ++kept;
    }
    __syncthreads();
  }
  if (lane == 0) counts[b] = kept;
}

}  // namespace

extern "C" int64_t ds2_yolo_postprocess_work_bytes(int32_t nb, int32_t N) {
  return (nb > 0 && N > 0) ? (int64_t)nb * (int64_t)yolo_work_per_image(N) : 0;
}

extern "C" int ds2_yolo_postprocess(const float* pred, int32_t nb, int32_t nc, int32_t N, float conf_thres, float iou_thres,
                                    int32_t max_det, const float* scale5, float* dets, int32_t* counts, void* work,
                                    int64_t work_bytes, void* stream) {
  DS2_REQUIRE(pred && dets && counts && work, "yolo_postprocess: null pointer");
  DS2_REQUIRE(nb > 0 && nc > 0 && N > 0 && max_det > 0 && nb <= 65535, "yolo_postprocess: bad sizes nb=%d nc=%d N=%d max_det=%d", nb, nc, N, max_det);
  DS2_REQUIRE(work_bytes >= ds2_yolo_postprocess_work_bytes(nb, N), "yolo_postprocess: work buffer too small (%lld < %lld bytes)",
              (long long)work_bytes, (long long)ds2_yolo_postprocess_work_bytes(nb, N));
  hipStream_t st = (hipStream_t)stream;
  unsigned char* wk = reinterpret_cast<unsigned char*>(work);
  hipLaunchKernelGGL(k_yolo_score, dim3((N + 255) / 256, nb), dim3(256), 0, st, pred, nc, N, wk);
  hipLaunchKernelGGL(k_yolo_compact, dim3(nb), dim3(1024), 0, st, N, conf_thres, wk);
  const int cap = N < YCAP ? N : YCAP;
  hipLaunchKernelGGL(k_yolo_rank, dim3((cap + 255) / 256, nb), dim3(256), 0, st, pred, nc, N, wk);
  const int nwb = (cap + 63) / 64;
  hipLaunchKernelGGL(k_yolo_mask, dim3(nwb, nwb, nb), dim3(64), 0, st, N, iou_thres, wk);
  hipLaunchKernelGGL(k_yolo_sweep, dim3(nb), dim3(64), 0, st, N, max_det, scale5, dets, counts, wk);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
