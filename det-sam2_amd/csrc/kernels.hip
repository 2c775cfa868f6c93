// HBM-bound and glue kernels of the Det-SAM2 hot path (everything that is not a GEMM or an
// attention).  fp32 token-major activations; coalescing along the channel dimension; 64-lane wave
// reductions via __shfl_xor.  Each kernel cites the reference code it reproduces.
#include "kernels.h"
#include <hip/hip_fp16.h>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {  // torch .to(bfloat16): round-to-nearest-even
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float half_bits_to_f32(uint16_t h) {
  __half_raw r;
  r.x = h;
  return __half2float(__half(r));
}

// ------------------------------------------------------------------ LayerNorm (rows x C), one wave / row
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ y, int ldy,
                                                   int rows, int C, float eps, int act) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(v) / (float)C + eps);
  float* yr = y + (size_t)row * ldy;
  for (int c = lane; c < C; c += 64) yr[c] = ds2_act((xr[c] - mean) * rstd * w[c] + b[c], act);
}

// LayerNorm whose consumer is a bf16x3 GEMM: emits the two bf16 planes of the result (x = hi + lo) instead of
// fp32 (gemm_split.hip operand format; columns C..ldp are zero).  hi/lo are viewed as packed pairs.
__device__ __forceinline__ unsigned ln_cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__global__ __launch_bounds__(256) void k_layernorm_split(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         const float* __restrict__ b, unsigned* __restrict__ hi,
                                                         unsigned* __restrict__ lo, int ldp, int rows, int C, float eps,
                                                         int act) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(v) / (float)C + eps);
  unsigned* hr = hi + (size_t)row * (ldp / 2);
  unsigned* lr = lo + (size_t)row * (ldp / 2);
  for (int c2 = lane; c2 < ldp / 2; c2 += 64) {
    const int c = 2 * c2;
    const float y0 = c < C ? ds2_act((xr[c] - mean) * rstd * w[c] + b[c], act) : 0.f;
    const float y1 = c + 1 < C ? ds2_act((xr[c + 1] - mean) * rstd * w[c + 1] + b[c + 1], act) : 0.f;
    const unsigned h = ln_cvt_pk_bf16(y0, y1);
    hr[c2] = h;
    lr[c2] = ln_cvt_pk_bf16(y0 - __uint_as_float(h << 16), y1 - __uint_as_float(h & 0xffff0000u));
  }
}

// out = a + alpha*b (row-broadcast b) emitted directly as bf16 planes (cross-attention key input memory + pos)
__global__ void k_add_bcast_split(const float* a, int lda, const float* b, int ldb, int b_mod, float alpha, uint2* hi,
                                  uint2* lo, int ldp, int rows, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = ldp / 4;
  if (i >= (size_t)rows * q) return;
  const int c4 = (int)(i % q);
  const size_t r = i / q;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < C) {
    const size_t rb = b_mod > 0 ? r % b_mod : r;
    const float4 x = *reinterpret_cast<const float4*>(a + r * lda + c4 * 4);
    const float4 y = *reinterpret_cast<const float4*>(b + rb * ldb + c4 * 4);
    v = make_float4(x.x + alpha * y.x, x.y + alpha * y.y, x.z + alpha * y.z, x.w + alpha * y.w);
  }
  uint2 h, l;
  h.x = ln_cvt_pk_bf16(v.x, v.y);
  h.y = ln_cvt_pk_bf16(v.z, v.w);
  l.x = ln_cvt_pk_bf16(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
  l.y = ln_cvt_pk_bf16(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
  hi[i] = h;
  lo[i] = l;
}

// Vectorised LayerNorm: one wave per row, the whole row held in registers (NV float4 per lane: C <= 256 NV), ONE
// 16-byte-per-lane read of x, two-pass statistics in registers (same arithmetic as k_layernorm: mean, then the sum of
// squared deviations), 16-byte fp32 stores or 8-byte plane stores.  HBM-bound: 4 B read + 4 B written per element.
template <int NV, bool SPLIT>
__global__ __launch_bounds__(256) void k_layernorm_vec(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ y, int ldy,
                                                       uint2* __restrict__ hi, uint2* __restrict__ lo, int ldp, int rows, int C,
                                                       float eps, int act, const float* __restrict__ add = nullptr,
                                                       float* __restrict__ y2 = nullptr, int add_mod = 0,
                                                       uint2* __restrict__ hi0 = nullptr, uint2* __restrict__ lo0 = nullptr, int mx = 0) {
  // add (row stride ldy, row index modulo add_mod when > 0): a second result y + add - as fp32 y2 (!SPLIT: the "queries +
  // query_pe" of the two-way transformer) or as THE operand planes (SPLIT: "keys + key_pe"; y, when given, still gets LN(x))
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    v[i] = c < C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (4 * (lane + 64 * i) < C) {
      const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      const float4 w4 = *reinterpret_cast<const float4*>(w + c), b4 = *reinterpret_cast<const float4*>(b + c);
      o.x = ds2_act((v[i].x - mean) * rstd * w4.x + b4.x, act);
      o.y = ds2_act((v[i].y - mean) * rstd * w4.y + b4.y, act);
      o.z = ds2_act((v[i].z - mean) * rstd * w4.z + b4.z, act);
      o.w = ds2_act((v[i].w - mean) * rstd * w4.w + b4.w, act);
    }
    if (SPLIT) {
      if (y && c < C) *reinterpret_cast<float4*>(y + (size_t)row * ldy + c) = o;
      if (hi0 && c < ldp) {   // (hi0 / lo0: the planes of LN(x) itself, next to those of LN(x) + add)
        uint2 h, l;
        h.x = ln_cvt_pk_bf16(o.x, o.y);
        h.y = ln_cvt_pk_bf16(o.z, o.w);
        l.x = ln_cvt_pk_bf16(o.x - __uint_as_float(h.x << 16), o.y - __uint_as_float(h.x & 0xffff0000u));
        l.y = ln_cvt_pk_bf16(o.z - __uint_as_float(h.y << 16), o.w - __uint_as_float(h.y & 0xffff0000u));
        hi0[(size_t)row * (ldp / 4) + (c >> 2)] = h;
        lo0[(size_t)row * (ldp / 4) + (c >> 2)] = l;
      }
      if (add && c < C) {
        const float4 a4 = *reinterpret_cast<const float4*>(add + (size_t)(add_mod > 0 ? row % add_mod : row) * ldy + c);
        o = make_float4(o.x + a4.x, o.y + a4.y, o.z + a4.z, o.w + a4.w);
      }
      if (c < ldp) {   // columns C..ldp are zero in both planes
        uint2 h, l;
        if (mx) {      // "MX" activation planes (common.h) for a consumer that multiplies in the two-MFMA-equivalent form
          ds2_mx_pair(o.x, o.y, false, h.x, l.x);
          ds2_mx_pair(o.z, o.w, false, h.y, l.y);
        } else {
        h.x = ln_cvt_pk_bf16(o.x, o.y);
        h.y = ln_cvt_pk_bf16(o.z, o.w);
        l.x = ln_cvt_pk_bf16(o.x - __uint_as_float(h.x << 16), o.y - __uint_as_float(h.x & 0xffff0000u));
        l.y = ln_cvt_pk_bf16(o.z - __uint_as_float(h.y << 16), o.w - __uint_as_float(h.y & 0xffff0000u));
        }
        hi[(size_t)row * (ldp / 4) + (c >> 2)] = h;
        lo[(size_t)row * (ldp / 4) + (c >> 2)] = l;
      }
    } else if (c < C) {
      *reinterpret_cast<float4*>(y + (size_t)row * ldy + c) = o;
      if (y2) {
        const float4 a4 = *reinterpret_cast<const float4*>(add + (size_t)row * ldy + c);
        *reinterpret_cast<float4*>(y2 + (size_t)row * ldy + c) = make_float4(o.x + a4.x, o.y + a4.y, o.z + a4.z, o.w + a4.w);
      }
    }
  }
}

// C = 64 (LayerNorm2d of the mask downsampler, 262 144 rows per 16 objects): k_layernorm_vec would keep 16 of a wave's 64 lanes
// busy - here a 16-lane group owns a row (4 rows per wave).  Same expressions; the 16-lane butterfly (xor 8, 4, 2, 1) adds the
// same values in the same order as wave_sum does over a wave whose other 48 lanes hold zeros: bit-identical.
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__global__ __launch_bounds__(256) void k_layernorm_c64(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ y, int ldy, int rows,
                                                       float eps, int act) {
  const int lane = threadIdx.x & 63, l = lane & 15;
  int row = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4);
  const bool live = row < rows;
  row = live ? row : rows - 1;
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + 4 * l);
  const float mean = group16_sum((v.x + v.y) + (v.z + v.w)) / 64.f;
  const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  const float rstd = 1.f / sqrtf(group16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / 64.f + eps);
  const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * l), b4 = *reinterpret_cast<const float4*>(b + 4 * l);
  float4 o;
  o.x = ds2_act((v.x - mean) * rstd * w4.x + b4.x, act);
  o.y = ds2_act((v.y - mean) * rstd * w4.y + b4.y, act);
  o.z = ds2_act((v.z - mean) * rstd * w4.z + b4.z, act);
  o.w = ds2_act((v.w - mean) * rstd * w4.w + b4.w, act);
  if (live) *reinterpret_cast<float4*>(y + (size_t)row * ldy + 4 * l) = o;
}

template <bool SPLIT>
static bool launch_layernorm_vec(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, void* hi, void* lo,
                                 int ldp, int rows, int C, float eps, int act, hipStream_t st, const float* add = nullptr,
                                 float* y2 = nullptr, int add_mod = 0, void* hi0 = nullptr, void* lo0 = nullptr, int mx = 0) {
  const int width = SPLIT ? ldp : C;
  const bool ok = C % 4 == 0 && ldx % 4 == 0 && ((SPLIT && !y && !add) || ldy % 4 == 0) && width <= 5 * 256 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(b) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(add) & 15) == 0 && (reinterpret_cast<uintptr_t>(y2) & 15) == 0;
  if (!ok) return false;
  const int nv = (width + 255) / 256;
  const dim3 grid(cdiv(rows, 4)), blk(256);
  uint2* h2 = reinterpret_cast<uint2*>(hi);
  uint2* l2 = reinterpret_cast<uint2*>(lo);
#define DS2_LN_CASE(N) \
  case N: hipLaunchKernelGGL((k_layernorm_vec<N, SPLIT>), grid, blk, 0, st, x, ldx, w, b, y, ldy, h2, l2, ldp, rows, C, eps, act, add, y2, add_mod, reinterpret_cast<uint2*>(hi0), reinterpret_cast<uint2*>(lo0), mx); break;
  switch (nv) {
    DS2_LN_CASE(1) DS2_LN_CASE(2) DS2_LN_CASE(3) DS2_LN_CASE(4) DS2_LN_CASE(5)
    default: return false;
  }
#undef DS2_LN_CASE
  return true;
}

// Three-layer MLP 256 -> 256 -> 256 -> n_out (ReLU, ReLU, last_act) on a handful of rows: the SAM hypernetwork / IoU /
// object-score heads and obj_ptr_proj (mask_decoder.py MLP :281-296, sam2_base.py:171-175) - 7 MLPs of 16 rows per
// tracked frame, i.e. 21 tile GEMMs + 21 operand splits of ~10 us each when done as GEMMs.  One block per row, the
// activations stay in LDS; a wave owns output features j and its lanes split the 256-long dot product (16-byte
// coalesced weight reads, butterfly reduction).  Exact fp32 FMA.
__device__ __forceinline__ void mlp3_256_row(float (&x)[2][256], int row, const float* __restrict__ A, int lda,
                                             const float* __restrict__ w0, const float* __restrict__ b0,
                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ w2, const float* __restrict__ b2, int n_out,
                                             float* __restrict__ out, int ldc, int last_act) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < 256) x[0][t] = A[(size_t)row * lda + t];
  __syncthreads();
#pragma unroll 1
  for (int layer = 0; layer < 3; ++layer) {
    const float* w = layer == 0 ? w0 : (layer == 1 ? w1 : w2);
    const float* b = layer == 0 ? b0 : (layer == 1 ? b1 : b2);
    const int nout = layer == 2 ? n_out : 256;
    const int act = layer == 2 ? last_act : DS2_ACT_RELU;
    const float4 xv = *reinterpret_cast<const float4*>(&x[layer & 1][4 * lane]);
    // 8 waves x 16 output features per pass: 16 independent 16-byte weight loads in flight per lane (a serial loop
    // pays the full L2 latency once per feature)
    for (int j0 = wave * 16; j0 < nout; j0 += 128) {
      float4 wv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int j = j0 + u < nout ? j0 + u : nout - 1;
        wv[u] = *reinterpret_cast<const float4*>(w + (size_t)j * 256 + 4 * lane);
      }
      float mine = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float su = wave_sum(fmaf(wv[u].x, xv.x, fmaf(wv[u].y, xv.y, fmaf(wv[u].z, xv.z, wv[u].w * xv.w))));
        mine = lane == u ? su : mine;
      }
      if (lane < 16 && j0 + lane < nout) {
        const int j = j0 + lane;
        const float v = ds2_act(mine + b[j], act);
        if (layer == 2) out[(size_t)row * ldc + j] = v;
        else x[(layer + 1) & 1][j] = v;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(512) void k_mlp3_256(const float* __restrict__ A, int lda, const float* __restrict__ w0,
                                                  const float* __restrict__ b0, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, int n_out, float* __restrict__ out, int ldc,
                                                  int last_act) {
  __shared__ __attribute__((aligned(16))) float x[2][256];
  mlp3_256_row(x, blockIdx.x, A, lda, w0, b0, w1, b1, w2, b2, n_out, out, ldc, last_act);
}
// Several independent MLPs over the same number of rows in ONE launch (blockIdx.y = the MLP): the six heads that read the
// decoder's output tokens (4 hypernetworks, IoU, object score) - same arithmetic per row as k_mlp3_256.
__global__ __launch_bounds__(512) void k_mlp3_256_batch(Mlp3Batch jb) {
  __shared__ __attribute__((aligned(16))) float x[2][256];
  const Mlp3Job& j = jb.job[blockIdx.y];
  mlp3_256_row(x, blockIdx.x, j.A, j.lda, j.w0, j.b0, j.w1, j.b1, j.w2, j.b2, j.n_out, j.out, j.ldc, j.last_act);
}

// ------------------------------------------------------------------ simple elementwise
__global__ void k_add_bcast(const float* a, int lda, const float* b, int ldb, int b_mod, float alpha, float* out,
                            int ldo, int rows, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * C) return;
  const int r = (int)(i / C), c = (int)(i - (size_t)r * C);
  const int rb = b_mod > 0 ? r % b_mod : r;
  out[(size_t)r * ldo + c] = a[(size_t)r * lda + c] + alpha * b[(size_t)rb * ldb + c];
}

__global__ void k_add_rowvec(const float* a, int lda, const float* vec, float* out, int ldo, int rows, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * C) return;
  const int r = (int)(i / C), c = (int)(i - (size_t)r * C);
  out[(size_t)r * ldo + c] = a[(size_t)r * lda + c] + vec[c];
}

// 2x2/stride-2 max pooling on a channels-last map (do_pool, hieradet.py:25-36).
__global__ void k_maxpool2x2(const float* in, int ld_in, float* out, int ld_out, int H, int W, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Ho = H / 2, Wo = W / 2;
  if (i >= (size_t)Ho * Wo * C) return;
  const int c = (int)(i % C);
  const int p = (int)(i / C), ox = p % Wo, oy = p / Wo;
  const float* p00 = in + ((size_t)(2 * oy) * W + 2 * ox) * ld_in + c;
  const float v = fmaxf(fmaxf(p00[0], p00[ld_in]), fmaxf(p00[(size_t)W * ld_in], p00[(size_t)W * ld_in + ld_in]));
  out[(size_t)p * ld_out + c] = v;
}

// FPN top-down: out = lateral + nearest-2x(coarse)   (image_encoder.py:112-125)
__global__ void k_up2_add(const float* lat, const float* coarse, float* out, int H, int W, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)H * W * C) return;
  const int c = (int)(i % C);
  const int p = (int)(i / C), x = p % W, y = p / W;
  out[i] = lat[i] + coarse[((size_t)(y / 2) * (W / 2) + x / 2) * C + c];
}

// Axial RoPE applied in place to 256-wide rows (apply_rotary_enc, position_encoding.py:196-220):
// consecutive pairs are complex numbers multiplied by cis[pos][pair]; pos = token index modulo the
// 64x64 grid (rope_k_repeat); the last (L - n_rope) rows of every batch item are left untouched
// (object-pointer tokens, transformer.py:335-341).
__global__ void k_rope(float* x, int ldx, const float* cis, int batch, int L, int n_rope, int grid_tokens) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one complex pair per thread
  if (i >= (size_t)batch * n_rope * 128) return;
  const int pr = (int)(i % 128);
  const size_t rt = i / 128;
  const int t = (int)(rt % n_rope), b = (int)(rt / n_rope);
  float2* px = reinterpret_cast<float2*>(x + ((size_t)b * L + t) * ldx) + pr;
  const float2 c = reinterpret_cast<const float2*>(cis)[(size_t)(t % grid_tokens) * 128 + pr];
  const float2 v = *px;
  *px = make_float2(v.x * c.x - v.y * c.y, v.x * c.y + v.y * c.x);
}

// PatchEmbed 7x7 / stride 4 / pad 3 as im2col (backbones/utils.py:69-96): frame fp16 [3,S,S]
// (the reference's stored frame, .float()'ed at sam2_video_predictor.py:1186) -> [(S/4)^2, 148] fp32,
// column = c*49 + ky*7 + kx (the flattened conv weight order), column 147 = 0.
__global__ void k_im2col_patch_f32(const float* frame, float* out, int S) {   // same, fp32 frame (module-level entry)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int G = S / 4;
  if (i >= (size_t)G * G * 148) return;
  const int col = (int)(i % 148);
  const int p = (int)(i / 148), ox = p % G, oy = p / G;
  float v = 0.f;
  if (col < 147) {
    const int c = col / 49, r = col % 49, ky = r / 7, kx = r % 7;
    const int iy = oy * 4 - 3 + ky, ix = ox * 4 - 3 + kx;
    if (iy >= 0 && iy < S && ix >= 0 && ix < S) v = frame[((size_t)c * S + iy) * S + ix];
  }
  out[i] = v;
}
__global__ void k_im2col_patch(const uint16_t* frame, float* out, int S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int G = S / 4;
  if (i >= (size_t)G * G * 148) return;
  const int col = (int)(i % 148);
  const int p = (int)(i / 148), ox = p % G, oy = p / G;
  float v = 0.f;
  if (col < 147) {
    const int c = col / 49, r = col % 49, ky = r / 7, kx = r % 7;
    const int iy = oy * 4 - 3 + ky, ix = ox * 4 - 3 + kx;
    if (iy >= 0 && iy < S && ix >= 0 && ix < S) v = half_bits_to_f32(frame[((size_t)c * S + iy) * S + ix]);
  }
  out[i] = v;
}

// Frame ingest (load_video_frames, misc.py:328-359) for S x S uint8 RGB input (identity resize):
// fp16(x/255) then fp16 in-place normalisation is a pure function of (channel, byte) -> 3x256 LUT.
__global__ void k_ingest_u8(const uint8_t* rgb, const uint16_t* lut, uint16_t* out, int n, int S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)S * S;
  if (i >= (size_t)n * per) return;
  const size_t f = i / per, p = i - f * per;
  const uint8_t* px = rgb + (f * per + p) * 3;
  uint16_t* o = out + f * 3 * per + p;
  o[0] = lut[px[0]];
  o[per] = lut[256 + px[1]];
  o[2 * per] = lut[512 + px[2]];
}

// Frame ingest with resize: cv2.resize(frame, (S, S)) for 8-bit input = OpenCV's fixed-point INTER_LINEAR (resize.cpp,
// HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>): taps/weights per
// destination column and row come precomputed from the host (tab: [xofs S | a0 S | a1 S | yofs S | b0 S | b1 S]), the
// kernel is pure integer arithmetic; the resized BYTE then goes through the same normalisation LUT as k_ingest_u8.
__global__ void k_ingest_resize_u8(const uint8_t* rgb, const int* tab, const uint16_t* lut, uint16_t* out, int n, int H, int W,
                                   int S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)S * S;
  if (i >= (size_t)n * per) return;
  const size_t f = i / per, p = i - f * per;
  const int dy = (int)(p / S), dx = (int)(p - (size_t)dy * S);
  const int x0 = tab[dx], a0 = tab[S + dx], a1 = tab[2 * S + dx];
  const int sy = tab[3 * S + dy], b0 = tab[4 * S + dy], b1 = tab[5 * S + dy];
  const int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
  const int y0 = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
  const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > H - 1 ? H - 1 : sy + 1);
  const uint8_t* r0 = rgb + (f * H + y0) * (size_t)W * 3;
  const uint8_t* r1 = rgb + (f * H + y1) * (size_t)W * 3;
  uint16_t* o = out + f * 3 * per + p;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int s0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
    const int s1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
    int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    o[c * per] = lut[c * 256 + v];
  }
}

__global__ void k_permute4(const float* in, float* out, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)d0 * d1 * d2 * d3;
  if (i >= n) return;
  int idx[4];
  size_t r = i;
  idx[3] = (int)(r % d3); r /= d3;
  idx[2] = (int)(r % d2); r /= d2;
  idx[1] = (int)(r % d1); r /= d1;
  idx[0] = (int)r;
  const int d[4] = {d0, d1, d2, d3};
  const int p[4] = {p0, p1, p2, p3};
  size_t o = 0;
  for (int k = 0; k < 4; ++k) o = o * d[p[k]] + idx[p[k]];
  out[o] = in[i];
}

__global__ void k_pad_cols(const float* in, int rows, int cols, float* out, int cols_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols_out) return;
  const int c = (int)(i % cols_out), r = (int)(i / cols_out);
  out[i] = c < cols ? in[(size_t)r * cols + c] : 0.f;
}

// ------------------------------------------------------------------ bilinear helpers (align_corners=False)
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_coef(int o, float scale, int in_size) {
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lerp l;
  l.i0 = (int)src;
  if (l.i0 > in_size - 1) l.i0 = in_size - 1;
  l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
  l.w1 = src - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}
__device__ __forceinline__ float bilerp(const float* img, int w, const Lerp& ly, const Lerp& lx) {
  const float t0 = lx.w0 * img[(size_t)ly.i0 * w + lx.i0] + lx.w1 * img[(size_t)ly.i0 * w + lx.i1];
  const float t1 = lx.w0 * img[(size_t)ly.i1 * w + lx.i0] + lx.w1 * img[(size_t)ly.i1 * w + lx.i1];
  return ly.w0 * t0 + ly.w1 * t1;
}

// low-res logits -> high-res mask input of the memory encoder:
// F.interpolate(bilinear) (sam2_base.py:355-360 / sam2_video_predictor.py:747) followed by
// sigmoid or binarise, *scale + bias (sam2_base.py:713-725).
__global__ void k_mask_upsample_transform(const float* low, float* high, int B, int hin, int hout, int mode,
                                          float scale, float bias) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * hout * hout) return;
  const int x = (int)(i % hout);
  const int y = (int)((i / hout) % hout);
  const int b = (int)(i / ((size_t)hout * hout));
  const float s = (float)hin / (float)hout;
  const float v = bilerp(low + (size_t)b * hin * hin, hin, lerp_coef(y, s, hin), lerp_coef(x, s, hin));
  float m;
  if (mode == 0) m = 1.f / (1.f + expf(-v));
  else if (mode == 1) m = v > 0.f ? 1.f : 0.f;
  else { high[i] = v; return; }
  high[i] = m * scale + bias;
}

// MaskDownSampler stage with tiny channel counts (memory_encoder.py:36-52): conv3x3/s2/p1 -> LayerNorm2d
// (eps 1e-6) -> GELU, one thread per output pixel, all COUT channels in registers.  in/out NHWC.
template <int CIN, int COUT>
__global__ void k_conv3x3s2_small(const float* in, const float* w, const float* bias, const float* lnw,
                                  const float* lnb, float* out, int B, int Hin) {
  __shared__ float ws[COUT * CIN * 9];
  for (int i = threadIdx.x; i < COUT * CIN * 9; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const int Ho = Hin / 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Ho * Ho) return;
  const int ox = (int)(i % Ho), oy = (int)((i / Ho) % Ho), b = (int)(i / ((size_t)Ho * Ho));
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = bias[o];
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy - 1 + ky;
    if (iy < 0 || iy >= Hin) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox - 1 + kx;
      if (ix < 0 || ix >= Hin) continue;
      const float* px = in + (((size_t)b * Hin + iy) * Hin + ix) * CIN;
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float v = px[c];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] += v * ws[(o * CIN + c) * 9 + ky * 3 + kx];
      }
    }
  }
  float mean = 0.f;
#pragma unroll
  for (int o = 0; o < COUT; ++o) mean += acc[o];
  mean /= (float)COUT;
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < COUT; ++o) { const float d = acc[o] - mean; var += d * d; }
  const float rstd = 1.f / sqrtf(var / (float)COUT + 1e-6f);
  float* po = out + i * COUT;
#pragma unroll
  for (int o = 0; o < COUT; ++o) po[o] = ds2_act((acc[o] - mean) * rstd * lnw[o] + lnb[o], DS2_ACT_GELU);
}

// k_mask_upsample_transform + k_conv3x3s2_small<1, 4> in one pass (memory_encoder.py:36-52 on sam2_base.py:355-360,713-725):
// a thread computes the nine high-res mask values under its 3x3 window from the low-res logits (bilinear taps + sigmoid /
// binarise, the same expressions in the same order, so the result is bit-identical to the two-kernel path), then conv ->
// LayerNorm2d -> GELU.  The 1024^2 mask (4 MiB per object written and read back) never exists in HBM; the 256^2 logits stay in
// L2.  Neighbouring threads share a window column: 2.25x the minimum arithmetic, all of it VALU in an HBM-write-bound kernel.
__global__ void k_mask_up_conv1(const float* __restrict__ low, int hin, int Hin, int mode, float scale, float mbias,
                                const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ lnw,
                                const float* __restrict__ lnb, float* __restrict__ out, int B) {
  constexpr int COUT = 4;
  __shared__ float ws[COUT * 9];
  for (int i = threadIdx.x; i < COUT * 9; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const int Ho = Hin / 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Ho * Ho) return;
  const int ox = (int)(i % Ho), oy = (int)((i / Ho) % Ho), b = (int)(i / ((size_t)Ho * Ho));
  const float* img = low + (size_t)b * hin * hin;
  const float s = (float)hin / (float)Hin;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = bias[o];
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy - 1 + ky;
    if (iy < 0 || iy >= Hin) continue;
    const Lerp ly = lerp_coef(iy, s, hin);
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox - 1 + kx;
      if (ix < 0 || ix >= Hin) continue;
      float v = bilerp(img, hin, ly, lerp_coef(ix, s, hin));
      if (mode == 0) v = (1.f / (1.f + expf(-v))) * scale + mbias;
      else if (mode == 1) v = (v > 0.f ? 1.f : 0.f) * scale + mbias;
#pragma unroll
      for (int o = 0; o < COUT; ++o) acc[o] += v * ws[o * 9 + ky * 3 + kx];
    }
  }
  float mean = 0.f;
#pragma unroll
  for (int o = 0; o < COUT; ++o) mean += acc[o];
  mean /= (float)COUT;
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < COUT; ++o) { const float d = acc[o] - mean; var += d * d; }
  const float rstd = 1.f / sqrtf(var / (float)COUT + 1e-6f);
  float4 r;
  r.x = ds2_act((acc[0] - mean) * rstd * lnw[0] + lnb[0], DS2_ACT_GELU);
  r.y = ds2_act((acc[1] - mean) * rstd * lnw[1] + lnb[1], DS2_ACT_GELU);
  r.z = ds2_act((acc[2] - mean) * rstd * lnw[2] + lnb[2], DS2_ACT_GELU);
  r.w = ds2_act((acc[3] - mean) * rstd * lnw[3] + lnb[3], DS2_ACT_GELU);
  *reinterpret_cast<float4*>(out + i * COUT) = r;
}

// im2col for conv3x3/s2/p1 on NHWC input; column = (ky*3+kx)*Cin + c (weights are repacked to match).
__global__ void k_im2col3x3s2(const float* in, float* out, int B, int Hin, int Cin) {
  const int Ho = Hin / 2, c4n = Cin / 4;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Ho * Ho * 9 * c4n) return;
  const int c4 = (int)(i % c4n);
  size_t r = i / c4n;
  const int tap = (int)(r % 9); r /= 9;
  const int ox = (int)(r % Ho), oy = (int)((r / Ho) % Ho), b = (int)(r / ((size_t)Ho * Ho));
  const int iy = 2 * oy - 1 + tap / 3, ix = 2 * ox - 1 + tap % 3;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (iy >= 0 && iy < Hin && ix >= 0 && ix < Hin)
    v = *reinterpret_cast<const float4*>(in + (((size_t)b * Hin + iy) * Hin + ix) * Cin + c4 * 4);
  *reinterpret_cast<float4*>(out + (((size_t)b * Ho + oy) * Ho + ox) * 9 * Cin + tap * Cin + c4 * 4) = v;
}

// The same im2col emitted directly as the consumer GEMM's bf16 operand planes [rows, ldp] (the split of k_split_rows, element for
// element): the GEMM then needs no operand-split pre-pass over the 151 MB column matrix.  Columns 9 Cin .. ldp (the K padding to a
// multiple of 32) are written as zeros by the threads of a tenth "tap".
__global__ void k_im2col3x3s2_split(const float* __restrict__ in, uint2* __restrict__ hi, uint2* __restrict__ lo, int ldp, int B, int Hin,
                                    int Cin) {
  const int Ho = Hin / 2, c4n = Cin / 4, ntap = ldp > 9 * Cin ? 10 : 9;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Ho * Ho * ntap * c4n) return;
  const int c4 = (int)(i % c4n);
  size_t r = i / c4n;
  const int tap = (int)(r % ntap); r /= ntap;
  const int col = tap * Cin + c4 * 4;
  if (col >= ldp) return;
  const int ox = (int)(r % Ho), oy = (int)((r / Ho) % Ho), b = (int)(r / ((size_t)Ho * Ho));
  const int iy = 2 * oy - 1 + tap / 3, ix = 2 * ox - 1 + tap % 3;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tap < 9 && iy >= 0 && iy < Hin && ix >= 0 && ix < Hin)
    v = *reinterpret_cast<const float4*>(in + (((size_t)b * Hin + iy) * Hin + ix) * Cin + c4 * 4);
  uint2 h, l;
  h.x = ln_cvt_pk_bf16(v.x, v.y);
  h.y = ln_cvt_pk_bf16(v.z, v.w);
  l.x = ln_cvt_pk_bf16(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
  l.y = ln_cvt_pk_bf16(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
  const size_t o = ((((size_t)b * Ho + oy) * Ho + ox) * ldp + col) / 4;
  hi[o] = h;
  lo[o] = l;
}

// CXBlock depth-wise 7x7 / pad 3 (memory_encoder.py:86-92), NHWC, weights repacked [49][C].
// One thread = 8 consecutive output pixels of one row for one channel: each input row segment (14 values) is loaded
// once and feeds all 8 outputs (4x fewer loads than one output per thread); lanes run along C (coalesced).
// 4 rows x 8 columns of outputs per thread: an input row is loaded once (14 values) and feeds up to four output rows, 4.4 loads
// per output instead of 12.25.  The row loop is NOT unrolled (unrolled, hipcc hoists all 140 loads: 218 registers): the weights
// of the (input row, output row) pair come from L1 inside it.  Per output the products are added in the order of k_dwconv7
// (input rows ascending = ky ascending, kx ascending, rows outside the map skipped), every one as a fused multiply-add (112
// v_pk_fma_f32); hipcc compiles k_dwconv7's loop with a few products left unfused (v_pk_mul + v_pk_add), so 17 % of the outputs
// differ from it in the last bit (tools/ubench/dwconv_cmp.hip: max |diff| 7.6e-6 at |x| ~ 5).  79 -> 52 us per launch.
__global__ __launch_bounds__(256) void k_dwconv7_t4(const float* __restrict__ in, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out, int B, int H, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int XB = H / 8, YB = H / 4;
  if (i >= (size_t)B * YB * XB * C) return;
  const int c = (int)(i % C);
  size_t r_ = i / C;
  const int xb = (int)(r_ % XB), yb = (int)((r_ / XB) % YB), b = (int)(r_ / ((size_t)XB * YB));
  const int x0 = xb * 8, y0 = yb * 4;
  float acc[4][8];
  const float bs = bias[c];
#pragma unroll
  for (int oy = 0; oy < 4; ++oy)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[oy][j] = bs;
#pragma unroll 1
  for (int r = 0; r < 10; ++r) {
    const int iy = y0 + r - 3;
    if (iy < 0 || iy >= H) continue;
    float v[14];
    const float* rp = in + (((size_t)b * H + iy) * H) * C + c;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const int ix = x0 + j - 3;
      v[j] = (ix >= 0 && ix < H) ? rp[(size_t)ix * C] : 0.f;
    }
#pragma unroll
    for (int oy = 0; oy < 4; ++oy) {
      const int ky = r - oy;          // block-uniform
      if (ky < 0 || ky > 6) continue;
      const float* wp = w + (size_t)(ky * 7) * C + c;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const float wv = wp[(size_t)kx * C];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[oy][j] += v[j + kx] * wv;
      }
    }
  }
#pragma unroll
  for (int oy = 0; oy < 4; ++oy)
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(((size_t)b * H + y0 + oy) * H + x0 + j) * C + c] = acc[oy][j];
}
__global__ void k_dwconv7(const float* in, const float* w, const float* bias, float* out, int B, int H, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int XB = H / 8;
  if (i >= (size_t)B * H * XB * C) return;
  const int c = (int)(i % C);
  size_t r = i / C;
  const int xb = (int)(r % XB), y = (int)((r / XB) % H), b = (int)(r / ((size_t)XB * H));
  const int x0 = xb * 8;
  float acc[8];
  const float bs = bias[c];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = bs;
  // (measured and not kept: the row loop fully unrolled with predicated loads instead of the skip - 79 -> 175 us per launch)
  for (int ky = 0; ky < 7; ++ky) {
    const int iy = y + ky - 3;
    if (iy < 0 || iy >= H) continue;
    float v[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const int ix = x0 + j - 3;
      v[j] = (ix >= 0 && ix < H) ? in[(((size_t)b * H + iy) * H + ix) * C + c] : 0.f;
    }
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const float wv = w[(ky * 7 + kx) * C + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j + kx] * wv;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) out[(((size_t)b * H + y) * H + x0 + j) * C + c] = acc[j];
}

// + (1 - is_obj) * no_obj_embed_spatial (sam2_base.py:735-741), then bf16 storage
// (sam2_video_predictor.py:1337,1396).
__global__ void k_memfeat_finish(const float* feat, const float* obj_logits, const float* no_obj_embed,
                                 uint16_t* out, int B, int tokens, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * tokens * C) return;
  const int c = (int)(i % C), b = (int)(i / ((size_t)tokens * C));
  float v = feat[i];
  if (!(obj_logits[b] > 0.f)) v += no_obj_embed[c];
  out[i] = f32_to_bf16_rne(v);
}

// ------------------------------------------------------------------ SAM heads
// tokens[b] = [obj_score_token, iou_token, mask_tokens(4)] ++ sparse point embeddings (P points + the
// padding point)  (mask_decoder.py:174-195; prompt_encoder.py:73-95; position_encoding.py:129-158).
// pad = 0: no padding point (PromptEncoder.forward with boxes given, prompt_encoder.py:158); sparse_in != nullptr: the sparse
// embeddings are the caller's [B, P, 256] (MaskDecoder.forward's own argument) and are copied behind the output tokens.
__global__ void k_prompt_tokens(const float* out_tokens6, const float* gauss, const float* point_emb4,
                                const float* not_a_point, const float* coords, const int* labels, int B, int P,
                                float image_size, float* tokens, int pad, const float* sparse_in) {
  const int T = 6 + P + pad;
  const int bt = blockIdx.x, b = bt / T, t = bt % T, j = threadIdx.x;  // 256 threads = 256 channels
  float v;
  if (t < 6) {
    v = out_tokens6[t * 256 + j];
  } else if (sparse_in) {
    v = sparse_in[((size_t)b * P + (t - 6)) * 256 + j];
  } else {
    const int p = t - 6;
    int lab = -1;
    float cx = 0.f, cy = 0.f;
    if (p < P) {
      lab = labels[b * P + p];
      cx = coords[(b * P + p) * 2 + 0] + 0.5f;
      cy = coords[(b * P + p) * 2 + 1] + 0.5f;
    }
    if (lab == -1) {
      v = not_a_point[j];
    } else {
      const float x = 2.f * (cx / image_size) - 1.f, y = 2.f * (cy / image_size) - 1.f;
      const int jj = j & 127;
      const float ang = 6.283185307179586f * (x * gauss[jj] + y * gauss[128 + jj]);
      v = (j < 128 ? sinf(ang) : cosf(ang));
      if (lab >= 0 && lab < 4) v += point_emb4[lab * 256 + j];
    }
  }
  tokens[(size_t)bt * 256 + j] = v;
}

// upscaled = GELU(LayerNorm2d(ConvT1(src) + feat_s1))  (mask_decoder.py:220-223): g1 holds the
// ConvTranspose2d(2x2,s2) as a GEMM result [B*4096, 4*64] with column (dy*2+dx)*64 + c.
// One wave per output pixel: the 64 channels are the 64 lanes.
__global__ __launch_bounds__(256) void k_upscale1(const float* g1, const float* feat_s1, const float* lnw,
                                                  const float* lnb, float* u1, int B, unsigned* hi, unsigned* lo) {
  const size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int c = threadIdx.x & 63;
  if (pix >= (size_t)B * 128 * 128) return;
  const int X = (int)(pix % 128), Y = (int)((pix / 128) % 128), b = (int)(pix / (128 * 128));
  const int tok = (Y / 2) * 64 + X / 2, sub = (Y & 1) * 2 + (X & 1);
  const float v = g1[((size_t)b * 4096 + tok) * 256 + sub * 64 + c] + feat_s1[((size_t)Y * 128 + X) * 64 + c];
  const float mean = wave_sum(v) / 64.f;
  const float d = v - mean;
  const float rstd = 1.f / sqrtf(wave_sum(d * d) / 64.f + 1e-6f);
  const float o = ds2_act(d * rstd * lnw[c] + lnb[c], DS2_ACT_GELU);
  if (hi) {   // operand planes [pixels, 64] instead of fp32: channels (c, c + 1) packed by the even lane
    const float on = __shfl_down(o, 1);
    if (!(c & 1)) {
      const unsigned h = ln_cvt_pk_bf16(o, on);
      hi[pix * 32 + (c >> 1)] = h;
      lo[pix * 32 + (c >> 1)] = ln_cvt_pk_bf16(o - __uint_as_float(h << 16), on - __uint_as_float(h & 0xffff0000u));
    }
  } else {
    u1[pix * 64 + c] = o;
  }
}

// masks[b,m] = hyper_in[b,m,:] . GELU(ConvT2(u1) + feat_s0)   (mask_decoder.py:224,235) without ever
// materialising the [B,32,256,256] upscaled embedding.  g2: [B*16384, 4*32].
__global__ __launch_bounds__(256) void k_upscale2_masks(const float* g2, const float* feat_s0, const float* hyper,
                                                        float* masks, int B) {
  __shared__ float hs[4 * 32];
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int b = (int)(((size_t)blockIdx.x * 256) / 65536);
  if (threadIdx.x < 128) hs[threadIdx.x] = hyper[b * 128 + threadIdx.x];
  __syncthreads();
  if (pix >= (size_t)B * 65536) return;
  const int X = (int)(pix % 256), Y = (int)((pix / 256) % 256);
  const int tok = (Y / 2) * 128 + X / 2, sub = (Y & 1) * 2 + (X & 1);
  const float4* pg = reinterpret_cast<const float4*>(g2 + ((size_t)b * 16384 + tok) * 128 + sub * 32);
  const float4* pf = reinterpret_cast<const float4*>(feat_s0 + ((size_t)Y * 256 + X) * 32);
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 g = pg[q], f = pf[q];
    const float u[4] = {ds2_act(g.x + f.x, DS2_ACT_GELU), ds2_act(g.y + f.y, DS2_ACT_GELU),
                        ds2_act(g.z + f.z, DS2_ACT_GELU), ds2_act(g.w + f.w, DS2_ACT_GELU)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = q * 4 + k;
      m0 += hs[c] * u[k]; m1 += hs[32 + c] * u[k]; m2 += hs[64 + c] * u[k]; m3 += hs[96 + c] * u[k];
    }
  }
  const size_t o = (size_t)b * 4 * 65536 + (size_t)Y * 256 + X;
  masks[o] = m0; masks[o + 65536] = m1; masks[o + 2 * 65536] = m2; masks[o + 3 * 65536] = m3;
}

__global__ void k_gather_rows(const float* in, int ld_in, int row_stride, int row_off, float* out, int ld_out, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  out[(size_t)b * ld_out + c] = in[((size_t)b * row_stride + row_off) * ld_in + c];
}

// Mask selection glue of MaskDecoder.forward + _forward_sam_heads (mask_decoder.py:140-155,246-296;
// sam2_base.py:343-370): multimask -> best of masks 1..3 by predicted IoU; otherwise mask 0 unless its
// stability score < thresh, then the best multimask.  Objectness gate -> NO_OBJ_SCORE.  One block / object.
__global__ __launch_bounds__(1024) void k_select_masks(const float* masks4, const float* iou4, const float* obj_logits,
                                                       const float* tokens_out, int tok_ld, int multimask, float delta,
                                                       float thresh, float* low_res, float* sel_token, float* iou_out) {
  __shared__ unsigned int cnt[2];
  __shared__ int s_idx;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* mb = masks4 + (size_t)b * 4 * 65536;
  if (tid < 2) cnt[tid] = 0;
  __syncthreads();
  if (!multimask) {
    unsigned int ci = 0, cu = 0;
    for (int i = tid; i < 65536; i += 1024) {
      const float v = mb[i];
      ci += v > delta;
      cu += v > -delta;
    }
    atomicAdd(&cnt[0], ci);
    atomicAdd(&cnt[1], cu);
  }
  __syncthreads();
  if (tid == 0) {
    const float* io = iou4 + b * 4;
    int best = 1;
    if (io[2] > io[best]) best = 2;
    if (io[3] > io[best]) best = 3;
    int idx = best;
    if (!multimask) {
      const float ai = (float)cnt[0], au = (float)cnt[1];
      const float stab = au > 0.f ? ai / au : 1.f;
      idx = stab >= thresh ? 0 : best;
    }
    s_idx = idx;
    if (iou_out) iou_out[b] = io[idx];
  }
  __syncthreads();
  const int idx = s_idx;
  const bool appearing = obj_logits[b] > 0.f;
  const float* src = mb + (size_t)idx * 65536;
  float* dst = low_res + (size_t)b * 65536;
  for (int i = tid; i < 65536; i += 1024) dst[i] = appearing ? src[i] : -1024.f;
  // SAM output token for the object pointer: mask token 0 unless multimask (then the best one).
  const int tok = 2 + (multimask ? idx : 0);
  if (tid < 256) sel_token[b * 256 + tid] = tokens_out[(size_t)b * tok_ld + tok * 256 + tid];
}

// obj_ptr = lam*ptr + (1-lam)*no_obj_ptr  (sam2_base.py:375-387, fixed_no_obj_ptr)
__global__ void k_ptr_gate(float* ptr, const float* obj_logits, const float* no_obj_ptr, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  const float lam = obj_logits[b] > 0.f ? 1.f : 0.f;
  ptr[i] = lam * ptr[i] + (1.f - lam) * no_obj_ptr[c];
}

// ------------------------------------------------------------------ mask prompts (F3: add_new_mask)
// F.interpolate(mode="bilinear", antialias=True, align_corners=False) as ATen computes it on the CPU
// (UpSampleKernel.cpp, HelperInterpLinear::aa + the separable loops: last dimension first): per output index
//   scale = in / out (float); support = scale >= 1 ? scale : 1; center = scale * (i + 0.5);
//   xmin = max(int(center - support + 0.5), 0); xsize = min(int(center + support + 0.5), in) - xmin;
//   w_j = tri((j + xmin - center + 0.5) * (scale >= 1 ? 1/scale : 1)), normalised by their sum; out = sum_j src[xmin+j] * w_j
// accumulated in tap order in fp32.  `along_w`: 1 = resize the last dimension (rows of `in` are [rows, n_in]), 0 = the
// first one of a [n_in, cols] image.  v = in * in_scale + in_bias is applied to the source first; thresh < inf turns
// the output into (out >= thresh) ? 1 : 0.
__global__ void k_resize_aa_1d(const float* in, float* out, int B, int n_in, int n_out, int other, int along_w, float in_scale,
                               float in_bias, float thresh) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * n_out * other) return;
  int o, r;            // o: output index along the resized dimension, r: index along the other one
  size_t b;
  if (along_w) { o = (int)(i % n_out); r = (int)((i / n_out) % other); b = i / ((size_t)n_out * other); }
  else { r = (int)(i % other); o = (int)((i / other) % n_out); b = i / ((size_t)n_out * other); }
  const float scale = (float)n_in / (float)n_out;
  const float support = scale >= 1.f ? scale : 1.f;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const float center = scale * ((float)o + 0.5f);
  int xmin = (int)(center - support + 0.5f);
  xmin = xmin > 0 ? xmin : 0;
  int xmax = (int)(center + support + 0.5f);
  xmax = xmax < n_in ? xmax : n_in;
  const int xsize = xmax - xmin;
  float total = 0.f;
  for (int j = 0; j < xsize; ++j) {
    float x = fabsf(((float)(j + xmin) - center + 0.5f) * invscale);
    total += x < 1.f ? 1.f - x : 0.f;
  }
  const float* src = along_w ? in + (b * other + r) * (size_t)n_in : in + b * (size_t)n_in * other + r;
  const size_t stride = along_w ? 1 : (size_t)other;
  float t = 0.f;
  for (int j = 0; j < xsize; ++j) {
    float x = fabsf(((float)(j + xmin) - center + 0.5f) * invscale);
    float w = x < 1.f ? 1.f - x : 0.f;
    if (total != 0.f) w = w / total;
    const float v = src[(size_t)(xmin + j) * stride] * in_scale + in_bias;
    t = j == 0 ? __fmul_rn(v, w) : __fadd_rn(t, __fmul_rn(v, w));
  }
  if (thresh < INFINITY) t = t >= thresh ? 1.f : 0.f;
  out[i] = t;
}

// mask_downsample (Conv2d(1, 1, kernel 4, stride 4), sam2_base.py:180-183) of a 0/1 mask [B,S,S] -> [B,S/4,S/4], and
// is_obj_appearing = any(mask > 0) -> object_score_logits = +-10 (sam2_base.py:436-440).  any[] must be zeroed before.
__global__ void k_mask_downsample4(const float* mask, const float* w16, const float* bias, float* out, int* any, int B, int S) {
  const int So = S / 4;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * So * So) return;
  const int x = (int)(i % So), y = (int)((i / So) % So);
  const size_t b = i / ((size_t)So * So);
  const float* m = mask + (b * S + 4 * y) * (size_t)S + 4 * x;
  float acc = 0.f;
  bool pos = false;
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      const float v = m[(size_t)ky * S + kx];
      acc += v * w16[ky * 4 + kx];
      pos |= v > 0.f;
    }
  out[i] = acc + bias[0];
  if (pos) atomicOr(any + b, 1);
}
__global__ void k_any_to_logits(const int* any, float* obj_logits, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) obj_logits[b] = any[b] ? 10.f : -10.f;
}

// ------------------------------------------------------------------ memory bank assembly (sam2_base.py:565-648)
__global__ void k_bank_mem(BankArgs a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B * n_mem * tokens * 16 (float4 granules)
  const size_t per_e = (size_t)a.tokens * 16;
  if (i >= (size_t)a.B * a.n_mem * per_e) return;
  const int c4 = (int)(i % 16);
  size_t r = i / 16;
  const int t = (int)(r % a.tokens); r /= a.tokens;
  const int e = (int)(r % a.n_mem), b = (int)(r / a.n_mem);
  const int Nk = a.Nk;
  const uint16_t* src = a.feats[e] + ((size_t)b * a.tokens + t) * 64 + c4 * 4;
  const ushort4 h = *reinterpret_cast<const ushort4*>(src);
  const float4 m = make_float4(bf16_bits_to_f32(h.x), bf16_bits_to_f32(h.y), bf16_bits_to_f32(h.z), bf16_bits_to_f32(h.w));
  const float4 pe = *reinterpret_cast<const float4*>(a.maskmem_pos + (size_t)t * 64 + c4 * 4);
  const float4 tp = *reinterpret_cast<const float4*>(a.tpos_enc + (size_t)a.tpos_row[e] * 64 + c4 * 4);
  const size_t o = ((size_t)b * Nk + (size_t)(a.e0 + e) * a.tokens + t) * 64 + c4 * 4;
  *reinterpret_cast<float4*>(a.mem + o) = m;
  *reinterpret_cast<float4*>(a.mem_pos + o) = make_float4(pe.x + tp.x, pe.y + tp.y, pe.z + tp.z, pe.w + tp.w);
}

// pointer tokens: split each 256-d pointer into 4 tokens of 64; pos = obj_ptr_tpos_proj(sine_pe(pos))
// (get_1d_sine_pe, sam2_utils.py:69-79).  One block (64 threads) per (pointer entry).
__global__ __launch_bounds__(64) void k_bank_ptr(BankArgs a, const float* dim_t) {
  __shared__ float pe[256];
  const int e = blockIdx.x, c = threadIdx.x;
  const float pos = a.ptr_pos[e];
  for (int k = c; k < 128; k += 64) {
    const float v = pos / dim_t[k];
    pe[k] = sinf(v);
    pe[128 + k] = cosf(v);
  }
  __syncthreads();
  float tp = 0.f;
  for (int k = 0; k < 256; ++k) tp += pe[k] * a.tpos_w[c * 256 + k];
  tp += a.tpos_b[c];
  const int Nk = a.Nk;
  for (int b = 0; b < a.B; ++b)
    for (int j = 0; j < 4; ++j) {
      const float m = a.ptrs[e][(size_t)b * 256 + j * 64 + c];
      const size_t o = ((size_t)b * Nk + (size_t)a.n_mem_total * a.tokens + (a.p0 + e) * 4 + j) * 64 + c;
      a.mem[o] = m;
      a.mem_pos[o] = tp;
    }
}

// The bank straight to the operands of the memory cross-attention (mode bf16x3k with the assembly attention, model.hip
// memory_attention_impl): the key input kin = memory + memory_pos as bf16 operand planes of the k_proj GEMM - the fp32 `memory` and
// `memory_pos` [B, Nk, 64] tensors (2 x 118 MB at 16 objects and 7 frames) are never written nor read back.  Same expressions as
// k_bank_mem followed by k_add_bcast_split: pos = pe + tp, then m + pos, then the bf16 split (bit-identical planes).
__global__ void k_bank_kin(BankArgs a, uint2* hi, uint2* lo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B * n_mem * tokens * 16 (4-column granules)
  const size_t per_e = (size_t)a.tokens * 16;
  if (i >= (size_t)a.B * a.n_mem * per_e) return;
  const int c4 = (int)(i % 16);
  size_t r = i / 16;
  const int t = (int)(r % a.tokens); r /= a.tokens;
  const int e = (int)(r % a.n_mem), b = (int)(r / a.n_mem);
  const ushort4 h4 = *reinterpret_cast<const ushort4*>(a.feats[e] + ((size_t)b * a.tokens + t) * 64 + c4 * 4);
  const float4 m = make_float4(bf16_bits_to_f32(h4.x), bf16_bits_to_f32(h4.y), bf16_bits_to_f32(h4.z), bf16_bits_to_f32(h4.w));
  const float4 pe = *reinterpret_cast<const float4*>(a.maskmem_pos + (size_t)t * 64 + c4 * 4);
  const float4 tp = *reinterpret_cast<const float4*>(a.tpos_enc + (size_t)a.tpos_row[e] * 64 + c4 * 4);
  const float4 pos = make_float4(pe.x + tp.x, pe.y + tp.y, pe.z + tp.z, pe.w + tp.w);
  const float4 v = make_float4(m.x + 1.0f * pos.x, m.y + 1.0f * pos.y, m.z + 1.0f * pos.z, m.w + 1.0f * pos.w);
  uint2 h, l;
  h.x = ln_cvt_pk_bf16(v.x, v.y);
  h.y = ln_cvt_pk_bf16(v.z, v.w);
  l.x = ln_cvt_pk_bf16(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
  l.y = ln_cvt_pk_bf16(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
  const size_t o = ((size_t)b * a.Nk + (size_t)(a.e0 + e) * a.tokens + t) * 16 + c4;
  hi[o] = h;
  lo[o] = l;
}
// pointer tokens of the same form: kin planes rows + their slots in the V^T tiles of the assembly attention (vt32: [b][tile][dv 64][slot
// 32] fp16, slot = vt_slot[key & 31]; the tiles were zeroed by the caller - keys past the bank's end are zero as k_vt_pack32 leaves them)
__global__ __launch_bounds__(64) void k_bank_ptr_planes(BankArgs a, const float* dim_t, unsigned short* hi, unsigned short* lo,
                                                        unsigned short* vt, int ntile, const unsigned char* vt_slot) {
  __shared__ float pe[256];
  const int e = blockIdx.x, c = threadIdx.x;
  const float pos = a.ptr_pos[e];
  for (int k = c; k < 128; k += 64) {
    const float v = pos / dim_t[k];
    pe[k] = sinf(v);
    pe[128 + k] = cosf(v);
  }
  __syncthreads();
  float tp = 0.f;
  for (int k = 0; k < 256; ++k) tp += pe[k] * a.tpos_w[c * 256 + k];
  tp += a.tpos_b[c];
  for (int b = blockIdx.y; b < a.B; b += gridDim.y)      // (one block per (entry, object): the serial loop over the objects was 28 us of latency)
    for (int j = 0; j < 4; ++j) {
      const float m = a.ptrs[e][(size_t)b * 256 + j * 64 + c];
      const int row = a.n_mem_total * a.tokens + (a.p0 + e) * 4 + j;
      const float v = m + 1.0f * tp;
      const unsigned h = ln_cvt_pk_bf16(v, 0.f) & 0xffffu;
      const unsigned l = ln_cvt_pk_bf16(v - __uint_as_float(h << 16), 0.f) & 0xffffu;
      const size_t o = ((size_t)b * a.Nk + row) * 64 + c;
      hi[o] = (unsigned short)h;
      lo[o] = (unsigned short)l;
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      typedef float f2 __attribute__((ext_vector_type(2)));
      const unsigned f = __builtin_bit_cast(unsigned, __builtin_convertvector((f2{ds2_sat_f16(m), 0.f}), h2)) & 0xffffu;
      vt[(((size_t)b * ntile + row / 32) * 64 + c) * 32 + vt_slot[row & 31]] = (unsigned short)f;
    }
}

// ------------------------------------------------------------------ output (A15): bilinear 256^2 -> video
// resolution (sam2_video_predictor.py:618-642) fused with `> 0` and bit-packing (det_sam2_RT.py:396-399);
// packing order = numpy.packbits (MSB first).  One thread per 8 output pixels.
__global__ void k_mask_output(const float* low, int B, int hin, int Hv, int Wv, float* logits, uint8_t* packed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int wb = (Wv + 7) / 8;          // packed row pitch: numpy.packbits pads the last byte with zero bits
  if (i >= (size_t)B * Hv * wb) return;
  const int xb = (int)(i % wb), y = (int)((i / wb) % Hv), b = (int)(i / ((size_t)wb * Hv));
  const float sy = (float)hin / (float)Hv, sx = (float)hin / (float)Wv;
  const Lerp ly = lerp_coef(y, sy, hin);
  const float* img = low + (size_t)b * hin * hin;
  unsigned int bits = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = xb * 8 + k;
    if (x >= Wv) break;
    const float v = (hin == Hv && hin == Wv) ? img[(size_t)y * hin + x] : bilerp(img, hin, ly, lerp_coef(x, sx, hin));
    if (logits) logits[((size_t)b * Hv + y) * Wv + x] = v;
    bits |= (v > 0.f ? 1u : 0u) << (7 - k);
  }
  if (packed) packed[i] = (uint8_t)bits;
}

inline dim3 grid1(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }


// ---- PromptEncoder._embed_masks (prompt_encoder.py:97-100, mask_downscaling :60-68): mask logits [B,256,256] ->
// dense prompt embedding [B,64*64,256]: conv2x2 s2 (1->4) + LayerNorm2d + GELU, conv2x2 s2 (4->16) + LayerNorm2d +
// GELU, conv1x1 (16->256); fused with src = image_embedding + dense (mask_decoder.py:203).  One block per token.
struct MaskDownArgs {
  const float *w0, *b0, *ln1w, *ln1b, *w3, *b3, *ln4w, *ln4b, *w6, *b6;
};
__global__ __launch_bounds__(256) void k_mask_downscale_add(const float* mask, MaskDownArgs a, const float* src, int src_bcast,
                                                            float* keys) {
  __shared__ float in[16], v1[16], g1[16], v2[16], g2[16];
  const int tok = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int ty = tok >> 6, tx = tok & 63;
  if (t < 16) in[t] = mask[((size_t)b * 256 + ty * 4 + (t >> 2)) * 256 + tx * 4 + (t & 3)];
  __syncthreads();
  if (t < 16) {   // t = position p (py, px) * 4 + channel c
    const int p = t >> 2, c = t & 3, py = p >> 1, px = p & 1;
    float acc = a.b0[c];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) acc = fmaf(a.w0[c * 4 + dy * 2 + dx], in[(py * 2 + dy) * 4 + px * 2 + dx], acc);
    v1[t] = acc;
  }
  __syncthreads();
  if (t < 16) {   // LayerNorm2d over the 4 channels of the position (sam2_utils.py:150-162, eps 1e-6) + GELU
    const int p = t >> 2, c = t & 3;
    const float u = 0.25f * (v1[p * 4] + v1[p * 4 + 1] + v1[p * 4 + 2] + v1[p * 4 + 3]);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s2 += (v1[p * 4 + k] - u) * (v1[p * 4 + k] - u);
    const float x = (v1[t] - u) / sqrtf(0.25f * s2 + 1e-6f);
    g1[t] = ds2_act(a.ln1w[c] * x + a.ln1b[c], DS2_ACT_GELU);
  }
  __syncthreads();
  if (t < 16) {   // conv 4->16 over the 2x2 positions: weight [16][4][2][2]
    float acc = a.b3[t];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc = fmaf(a.w3[(t * 4 + c) * 4 + p], g1[p * 4 + c], acc);
    v2[t] = acc;
  }
  __syncthreads();
  if (t < 16) {
    float u = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) u += v2[k];
    u *= 0.0625f;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s2 += (v2[k] - u) * (v2[k] - u);
    const float x = (v2[t] - u) / sqrtf(0.0625f * s2 + 1e-6f);
    g2[t] = ds2_act(a.ln4w[t] * x + a.ln4b[t], DS2_ACT_GELU);
  }
  __syncthreads();
  float acc = a.b6[t];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc = fmaf(a.w6[t * 16 + c], g2[c], acc);
  const size_t o = ((size_t)b * 4096 + tok) * 256 + t;
  keys[o] = src[src_bcast ? (size_t)tok * 256 + t : o] + acc;
}

}  // namespace

// ====================================================================== launchers
int launch_layernorm(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int rows, int C,
                     float eps, int act, hipStream_t st) {
  DS2_REQUIRE(rows > 0 && C > 0, "layernorm: bad dims");
  if (C == 64 && ldx % 4 == 0 && ldy % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(b)) & 15) == 0) {
    hipLaunchKernelGGL(k_layernorm_c64, dim3(cdiv(rows, 16)), dim3(256), 0, st, x, ldx, w, b, y, ldy, rows, eps, act);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
  }
  if (launch_layernorm_vec<false>(x, ldx, w, b, y, ldy, nullptr, nullptr, 0, rows, C, eps, act, st)) {
    DS2_CHECK_LAUNCH();
    return DS2_OK;
  }
  hipLaunchKernelGGL(k_layernorm, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, ldx, w, b, y, ldy, rows, C, eps, act);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
// y = LayerNorm(x), y2 = y + add (all three with leading dimension ldy): one launch for "norm, then + positional encoding"
int launch_layernorm_add(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, const float* add, float* y2,
                         int rows, int C, float eps, hipStream_t st) {
  DS2_REQUIRE(rows > 0 && C > 0 && add && y2, "layernorm_add: bad argument");
  const bool aligned = ((reinterpret_cast<uintptr_t>(add) | reinterpret_cast<uintptr_t>(y2)) & 15) == 0;
  if (aligned && launch_layernorm_vec<false>(x, ldx, w, b, y, ldy, nullptr, nullptr, 0, rows, C, eps, DS2_ACT_NONE, st, add, y2)) {
    DS2_CHECK_LAUNCH();
    return DS2_OK;
  }
  const int rc = launch_layernorm(x, ldx, w, b, y, ldy, rows, C, eps, DS2_ACT_NONE, st);
  if (rc != DS2_OK) return rc;
  return launch_add_bcast(y, ldy, add, ldy, 0, 1.f, y2, ldy, rows, C, st);
}
// y = LayerNorm(x) (fp32, leading dimension ldy) AND the operand planes of y + add[row % add_mod] (add: row stride ldy)
// (hi0 / lo0, optional: the planes of y itself as well)
int launch_layernorm_add_split(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, const float* add,
                               int add_mod, void* hi, void* lo, int ldp, int rows, int C, float eps, hipStream_t st, void* hi0,
                               void* lo0) {
  DS2_REQUIRE(rows > 0 && C > 0 && ldp % 32 == 0 && ldp >= C && y && add && hi && lo && !hi0 == !lo0, "layernorm_add_split: bad argument");
  if (launch_layernorm_vec<true>(x, ldx, w, b, y, ldy, hi, lo, ldp, rows, C, eps, DS2_ACT_NONE, st, add, nullptr, add_mod, hi0, lo0)) {
    DS2_CHECK_LAUNCH();
    return DS2_OK;
  }
  int rc = launch_layernorm(x, ldx, w, b, y, ldy, rows, C, eps, DS2_ACT_NONE, st);
  if (rc != DS2_OK) return rc;
  if (hi0 && (rc = launch_split_rows(y, ldy, rows, C, hi0, lo0, ldp, st)) != DS2_OK) return rc;
  return launch_add_bcast_split(y, ldy, add, ldy, add_mod, 1.f, hi, lo, ldp, rows, C, st);
}
// SAM decoder entry: keys = src[(row % src_mod)] + vec (fp32 + operand planes) and the planes of keys + pe[row % pe_mod]
// (mask_decoder.py:203, transformer.py:196-197) in one pass; C = ldp = 256
__global__ __launch_bounds__(256) void k_sam_keys_init(const float* __restrict__ src, int src_mod, const float* __restrict__ vec,
                                                       const float* __restrict__ pe, int pe_mod, float* __restrict__ keys,
                                                       uint2* __restrict__ khi, uint2* __restrict__ klo, uint2* __restrict__ phi,
                                                       uint2* __restrict__ plo, int rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4
  if (i >= (size_t)rows * 64) return;
  const int row = (int)(i >> 6), c = (int)(i & 63) * 4;
  const float4 s4 = *reinterpret_cast<const float4*>(src + (size_t)(src_mod > 0 ? row % src_mod : row) * 256 + c);
  const float4 v4 = *reinterpret_cast<const float4*>(vec + c);
  const float4 p4 = *reinterpret_cast<const float4*>(pe + (size_t)(pe_mod > 0 ? row % pe_mod : row) * 256 + c);
  const float4 k = make_float4(s4.x + v4.x, s4.y + v4.y, s4.z + v4.z, s4.w + v4.w);
  *reinterpret_cast<float4*>(keys + (size_t)row * 256 + c) = k;
  auto split4 = [](const float4& o, uint2& h, uint2& l) {
    h.x = ln_cvt_pk_bf16(o.x, o.y);
    h.y = ln_cvt_pk_bf16(o.z, o.w);
    l.x = ln_cvt_pk_bf16(o.x - __uint_as_float(h.x << 16), o.y - __uint_as_float(h.x & 0xffff0000u));
    l.y = ln_cvt_pk_bf16(o.z - __uint_as_float(h.y << 16), o.w - __uint_as_float(h.y & 0xffff0000u));
  };
  uint2 h, l;
  split4(k, h, l);
  khi[i] = h; klo[i] = l;
  split4(make_float4(k.x + p4.x, k.y + p4.y, k.z + p4.z, k.w + p4.w), h, l);
  phi[i] = h; plo[i] = l;
}
int launch_sam_keys_init(const float* src, int src_mod, const float* vec, const float* pe, int pe_mod, float* keys, void* khi,
                         void* klo, void* phi, void* plo, int rows, hipStream_t st) {
  DS2_REQUIRE(rows > 0 && src && vec && pe && keys && khi && klo && phi && plo, "sam_keys_init: bad argument");
  hipLaunchKernelGGL(k_sam_keys_init, grid1((size_t)rows * 64), dim3(256), 0, st, src, src_mod, vec, pe, pe_mod, keys,
                     reinterpret_cast<uint2*>(khi), reinterpret_cast<uint2*>(klo), reinterpret_cast<uint2*>(phi),
                     reinterpret_cast<uint2*>(plo), rows);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_layernorm_split(const float* x, int ldx, const float* w, const float* b, void* hi, void* lo, int ldp, int rows,
                           int C, float eps, int act, hipStream_t st, int mx) {
  DS2_REQUIRE(rows > 0 && C > 0 && ldp % 32 == 0 && ldp >= C, "layernorm_split: bad dims");
  if (launch_layernorm_vec<true>(x, ldx, w, b, nullptr, 0, hi, lo, ldp, rows, C, eps, act, st, nullptr, nullptr, 0, nullptr, nullptr, mx)) {
    DS2_CHECK_LAUNCH();
    return DS2_OK;
  }
  DS2_REQUIRE(!mx, "layernorm_split: MX planes need the vectorised kernel (C %% 4 == 0, 16-byte aligned operands, width <= 1280)");
  hipLaunchKernelGGL(k_layernorm_split, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, ldx, w, b, reinterpret_cast<unsigned*>(hi),
                     reinterpret_cast<unsigned*>(lo), ldp, rows, C, eps, act);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_add_bcast_split(const float* a, int lda, const float* b, int ldb, int b_mod, float alpha, void* hi, void* lo,
                           int ldp, int rows, int C, hipStream_t st) {
  DS2_REQUIRE(ldp % 32 == 0 && C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "add_bcast_split: bad dims");
  hipLaunchKernelGGL(k_add_bcast_split, grid1((size_t)rows * (ldp / 4)), dim3(256), 0, st, a, lda, b, ldb, b_mod, alpha,
                     reinterpret_cast<uint2*>(hi), reinterpret_cast<uint2*>(lo), ldp, rows, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_add_bcast(const float* a, int lda, const float* b, int ldb, int b_mod, float alpha, float* out, int ldo,
                     int rows, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_add_bcast, grid1((size_t)rows * C), dim3(256), 0, st, a, lda, b, ldb, b_mod, alpha, out, ldo, rows, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
// out[b][i] = x[i], b < B (n % 4 == 0): the shared layer-0 result of the memory attention replicated per object
__global__ void k_bcast_rows(const f32x4* __restrict__ x, f32x4* __restrict__ out, int n4, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = x[i];
  for (int b = 0; b < B; ++b) __builtin_nontemporal_store(v, out + (size_t)b * n4 + i);
}
int launch_bcast_rows(const float* x, float* out, int n, int B, hipStream_t st) {
  DS2_REQUIRE(n % 4 == 0, "bcast_rows: n must be a multiple of 4");
  hipLaunchKernelGGL(k_bcast_rows, dim3((n / 4 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x),
                     reinterpret_cast<f32x4*>(out), n / 4, B);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_add_rowvec(const float* a, int lda, const float* vec, float* out, int ldo, int rows, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_add_rowvec, grid1((size_t)rows * C), dim3(256), 0, st, a, lda, vec, out, ldo, rows, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_maxpool2x2(const float* in, int ld_in, float* out, int ld_out, int H, int W, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_maxpool2x2, grid1((size_t)(H / 2) * (W / 2) * C), dim3(256), 0, st, in, ld_in, out, ld_out, H, W, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_up2_add(const float* lat, const float* coarse, float* out, int H, int W, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_up2_add, grid1((size_t)H * W * C), dim3(256), 0, st, lat, coarse, out, H, W, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_rope(float* x, int ldx, const float* cis, int batch, int L, int n_rope, int grid_tokens, hipStream_t st) {
  if (n_rope <= 0) return DS2_OK;
  hipLaunchKernelGGL(k_rope, grid1((size_t)batch * n_rope * 128), dim3(256), 0, st, x, ldx, cis, batch, L, n_rope, grid_tokens);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_im2col_patch_f32(const float* frame_f32, float* out, int S, hipStream_t st) {
  hipLaunchKernelGGL(k_im2col_patch_f32, grid1((size_t)(S / 4) * (S / 4) * 148), dim3(256), 0, st, frame_f32, out, S);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_im2col_patch(const uint16_t* frame_f16, float* out, int S, hipStream_t st) {
  hipLaunchKernelGGL(k_im2col_patch, grid1((size_t)(S / 4) * (S / 4) * 148), dim3(256), 0, st, frame_f16, out, S);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_ingest_u8(const uint8_t* rgb, const uint16_t* lut, uint16_t* out, int n, int S, hipStream_t st) {
  hipLaunchKernelGGL(k_ingest_u8, grid1((size_t)n * S * S), dim3(256), 0, st, rgb, lut, out, n, S);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_ingest_resize_u8(const uint8_t* rgb, const int* tab, const uint16_t* lut, uint16_t* out, int n, int H, int W, int S,
                            hipStream_t st) {
  hipLaunchKernelGGL(k_ingest_resize_u8, grid1((size_t)n * S * S), dim3(256), 0, st, rgb, tab, lut, out, n, H, W, S);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_permute4(const float* in, float* out, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3,
                    hipStream_t st) {
  hipLaunchKernelGGL(k_permute4, grid1((size_t)d0 * d1 * d2 * d3), dim3(256), 0, st, in, out, d0, d1, d2, d3, p0, p1, p2, p3);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_pad_cols(const float* in, int rows, int cols, float* out, int cols_out, hipStream_t st) {
  hipLaunchKernelGGL(k_pad_cols, grid1((size_t)rows * cols_out), dim3(256), 0, st, in, rows, cols, out, cols_out);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_mask_upsample_transform(const float* low, float* high, int B, int hin, int hout, int mode, float scale,
                                   float bias, hipStream_t st) {
  hipLaunchKernelGGL(k_mask_upsample_transform, grid1((size_t)B * hout * hout), dim3(256), 0, st, low, high, B, hin, hout,
                     mode, scale, bias);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_mask_up_conv1(const float* low, int hin, int Hin, int mode, float scale, float mbias, const float* w, const float* bias,
                         const float* lnw, const float* lnb, float* out, int B, hipStream_t st) {
  const size_t n = (size_t)B * (Hin / 2) * (Hin / 2);
  hipLaunchKernelGGL(k_mask_up_conv1, grid1(n), dim3(256), 0, st, low, hin, Hin, mode, scale, mbias, w, bias, lnw, lnb, out, B);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_conv3x3s2_small(const float* in, const float* w, const float* bias, const float* lnw, const float* lnb,
                           float* out, int B, int Hin, int Cin, int Cout, hipStream_t st) {
  const size_t n = (size_t)B * (Hin / 2) * (Hin / 2);
  if (Cin == 1 && Cout == 4)
    hipLaunchKernelGGL((k_conv3x3s2_small<1, 4>), grid1(n), dim3(256), 0, st, in, w, bias, lnw, lnb, out, B, Hin);
  else if (Cin == 4 && Cout == 16)
    hipLaunchKernelGGL((k_conv3x3s2_small<4, 16>), grid1(n), dim3(256), 0, st, in, w, bias, lnw, lnb, out, B, Hin);
  else {
    ds2_set_error("conv3x3s2_small: unsupported channels %d->%d", Cin, Cout);
    return DS2_ERR_UNSUPPORTED;
  }
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_im2col3x3s2(const float* in, float* out, int B, int Hin, int Cin, hipStream_t st) {
  DS2_REQUIRE(Cin % 4 == 0, "im2col3x3s2: Cin must be a multiple of 4");
  hipLaunchKernelGGL(k_im2col3x3s2, grid1((size_t)B * (Hin / 2) * (Hin / 2) * 9 * (Cin / 4)), dim3(256), 0, st, in, out, B, Hin, Cin);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_im2col3x3s2_split(const float* in, void* hi, void* lo, int ldp, int B, int Hin, int Cin, hipStream_t st) {
  DS2_REQUIRE(Cin % 4 == 0 && ldp % 4 == 0 && ldp >= 9 * Cin && ldp - 9 * Cin <= Cin, "im2col3x3s2_split: bad plane pitch");
  const int ntap = ldp > 9 * Cin ? 10 : 9;
  hipLaunchKernelGGL(k_im2col3x3s2_split, grid1((size_t)B * (Hin / 2) * (Hin / 2) * ntap * (Cin / 4)), dim3(256), 0, st, in,
                     reinterpret_cast<uint2*>(hi), reinterpret_cast<uint2*>(lo), ldp, B, Hin, Cin);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_dwconv7(const float* in, const float* w49c, const float* bias, float* out, int B, int H, int C, hipStream_t st) {
  DS2_REQUIRE(H % 8 == 0, "dwconv7: H must be a multiple of 8");
  if (H % 4 == 0) {
    hipLaunchKernelGGL(k_dwconv7_t4, grid1((size_t)B * (H / 4) * (H / 8) * C), dim3(256), 0, st, in, w49c, bias, out, B, H, C);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
  }
  hipLaunchKernelGGL(k_dwconv7, grid1((size_t)B * H * (H / 8) * C), dim3(256), 0, st, in, w49c, bias, out, B, H, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_memfeat_finish(const float* feat, const float* obj_logits, const float* no_obj_embed, uint16_t* out_bf16,
                          int B, int tokens, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_memfeat_finish, grid1((size_t)B * tokens * C), dim3(256), 0, st, feat, obj_logits, no_obj_embed,
                     out_bf16, B, tokens, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_prompt_tokens(const float* out_tokens6, const float* gauss, const float* point_emb4, const float* not_a_point,
                         const float* coords, const int* labels, int B, int P, float image_size, float* tokens,
                         hipStream_t st, int pad, const float* sparse_in) {
  if (B * (6 + P + pad) <= 0) return DS2_OK;
  hipLaunchKernelGGL(k_prompt_tokens, dim3(B * (6 + P + pad)), dim3(256), 0, st, out_tokens6, gauss, point_emb4, not_a_point,
                     coords, labels, B, P, image_size, tokens, pad, sparse_in);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_upscale1(const float* g1, const float* feat_s1, const float* lnw, const float* lnb, float* u1, int B,
                    hipStream_t st, void* hi, void* lo) {
  DS2_REQUIRE(!hi == !lo, "upscale1: both planes or none");
  hipLaunchKernelGGL(k_upscale1, dim3((unsigned)((size_t)B * 128 * 128 / 4)), dim3(256), 0, st, g1, feat_s1, lnw, lnb, u1, B,
                     reinterpret_cast<unsigned*>(hi), reinterpret_cast<unsigned*>(lo));
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_upscale2_masks(const float* g2, const float* feat_s0, const float* hyper, float* masks, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_upscale2_masks, dim3((unsigned)((size_t)B * 65536 / 256)), dim3(256), 0, st, g2, feat_s0, hyper, masks, B);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_gather_rows(const float* in, int ld_in, int row_stride, int row_off, float* out, int ld_out, int B, int C,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_gather_rows, grid1((size_t)B * C), dim3(256), 0, st, in, ld_in, row_stride, row_off, out, ld_out, B, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_select_masks(const float* masks4, const float* iou4, const float* obj_logits, const float* tokens_out,
                        int tok_ld, int multimask, float delta, float thresh, float* low_res, float* sel_token,
                        float* iou_out, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_select_masks, dim3(B), dim3(1024), 0, st, masks4, iou4, obj_logits, tokens_out, tok_ld, multimask,
                     delta, thresh, low_res, sel_token, iou_out);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_ptr_gate(float* ptr, const float* obj_logits, const float* no_obj_ptr, int B, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_ptr_gate, grid1((size_t)B * C), dim3(256), 0, st, ptr, obj_logits, no_obj_ptr, B, C);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_resize_aa(const float* in, float* work, float* out, int B, int Hin, int Win, int Hout, int Wout, float in_scale,
                     float in_bias, float thresh, hipStream_t st) {
  // last dimension first, as ATen's separable CPU path; the affine map of the source is applied in the first pass only
  hipLaunchKernelGGL(k_resize_aa_1d, grid1((size_t)B * Hin * Wout), dim3(256), 0, st, in, work, B, Win, Wout, Hin, 1, in_scale,
                     in_bias, INFINITY);
  DS2_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_resize_aa_1d, grid1((size_t)B * Hout * Wout), dim3(256), 0, st, work, out, B, Hin, Hout, Wout, 0, 1.f, 0.f,
                     thresh);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_mask_downsample4(const float* mask, const float* w16, const float* bias, float* out, int* any, float* obj_logits,
                            int B, int S, hipStream_t st) {
  DS2_CHECK_HIP(hipMemsetAsync(any, 0, (size_t)B * sizeof(int), st));
  hipLaunchKernelGGL(k_mask_downsample4, grid1((size_t)B * (S / 4) * (S / 4)), dim3(256), 0, st, mask, w16, bias, out, any, B, S);
  DS2_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_any_to_logits, grid1((size_t)B), dim3(256), 0, st, any, obj_logits, B);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
int launch_bank_assemble(const BankArgs& a, hipStream_t st) {
  DS2_REQUIRE(a.n_mem >= 0 && a.n_mem <= DS2_MAX_MEM_ENTRIES && a.n_ptr >= 0 && a.n_ptr <= DS2_MAX_PTR_ENTRIES,
              "bank_assemble: too many entries (n_mem=%d n_ptr=%d)", a.n_mem, a.n_ptr);
  if (a.n_mem > 0) {
    hipLaunchKernelGGL(k_bank_mem, grid1((size_t)a.B * a.n_mem * a.tokens * 16), dim3(256), 0, st, a);
    DS2_CHECK_LAUNCH();
  }
  return DS2_OK;
}
int launch_bank_ptr(const BankArgs& a, const float* dim_t, hipStream_t st) {
  if (a.n_ptr > 0) {
    hipLaunchKernelGGL(k_bank_ptr, dim3(a.n_ptr), dim3(64), 0, st, a, dim_t);
    DS2_CHECK_LAUNCH();
  }
  return DS2_OK;
}
int launch_bank_kin(const BankArgs& a, void* hi, void* lo, hipStream_t st) {
  DS2_REQUIRE(a.n_mem >= 0 && a.n_mem <= DS2_MAX_MEM_ENTRIES, "bank_kin: too many entries (n_mem=%d)", a.n_mem);
  if (a.n_mem > 0) {
    hipLaunchKernelGGL(k_bank_kin, grid1((size_t)a.B * a.n_mem * a.tokens * 16), dim3(256), 0, st, a, reinterpret_cast<uint2*>(hi),
                       reinterpret_cast<uint2*>(lo));
    DS2_CHECK_LAUNCH();
  }
  return DS2_OK;
}
int launch_bank_ptr_planes(const BankArgs& a, const float* dim_t, void* hi, void* lo, void* vt32, int ntile, const unsigned char* vt_slot,
                           hipStream_t st) {
  DS2_REQUIRE(a.n_ptr >= 0 && a.n_ptr <= DS2_MAX_PTR_ENTRIES, "bank_ptr_planes: too many entries (n_ptr=%d)", a.n_ptr);
  if (a.n_ptr > 0) {
    hipLaunchKernelGGL(k_bank_ptr_planes, dim3(a.n_ptr, a.B), dim3(64), 0, st, a, dim_t, reinterpret_cast<unsigned short*>(hi),
                       reinterpret_cast<unsigned short*>(lo), reinterpret_cast<unsigned short*>(vt32), ntile, vt_slot);
    DS2_CHECK_LAUNCH();
  }
  return DS2_OK;
}
int launch_mask_output(const float* low, int B, int hin, int Hv, int Wv, float* logits, uint8_t* packed, hipStream_t st) {
  hipLaunchKernelGGL(k_mask_output, grid1((size_t)B * Hv * ((Wv + 7) / 8)), dim3(256), 0, st, low, B, hin, Hv, Wv, logits, packed);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

int launch_mask_downscale_add(const float* mask, const float* const* prm, const float* src, int src_bcast, float* keys, int B,
                              hipStream_t st) {
  MaskDownArgs a{prm[0], prm[1], prm[2], prm[3], prm[4], prm[5], prm[6], prm[7], prm[8], prm[9]};
  hipLaunchKernelGGL(k_mask_downscale_add, dim3(4096, B), dim3(256), 0, st, mask, a, src, src_bcast, keys);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

int launch_mlp3_256_batch(const Mlp3Batch& jb, int n_jobs, int rows, hipStream_t st) {
  DS2_REQUIRE(rows > 0 && n_jobs > 0 && n_jobs <= 8, "mlp3 batch: bad argument");
  for (int i = 0; i < n_jobs; ++i) {
    const Mlp3Job& j = jb.job[i];
    DS2_REQUIRE(j.A && j.out && j.n_out > 0 && j.n_out <= 256 && j.w0 && j.b0 && j.w1 && j.b1 && j.w2 && j.b2, "mlp3 batch: bad job %d", i);
  }
  hipLaunchKernelGGL(k_mlp3_256_batch, dim3(rows, n_jobs), dim3(512), 0, st, jb);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

int launch_mlp3_256(const float* A, int lda, const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                    const float* b2, int n_out, float* out, int ldc, int last_act, int rows, hipStream_t st) {
  DS2_REQUIRE(rows > 0 && n_out > 0 && n_out <= 256 && w0 && b0 && w1 && b1 && w2 && b2, "mlp3: bad argument");
  hipLaunchKernelGGL(k_mlp3_256, dim3(rows), dim3(512), 0, st, A, lda, w0, b0, w1, b1, w2, b2, n_out, out, ldc, last_act);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}
