// Flash attention (online softmax, exact-fp32 MFMA) for every attention on the Det-SAM2 hot path:
//   * Hiera windowed / q-pooled / global attention  (hieradet.py:57-82; backbones/utils.py:16-66)
//   * MemoryAttention self- and cross-attention      (transformer.py:312-363)  D=256, DV in {256,64}
//   * TwoWayTransformer token<->image attention      (transformer.py:239-284)  D in {32,16}
//
// wave64 design.  A block = 4 waves = 128 queries; wave w owns 32 queries and keeps them in
// registers as the MFMA *B* operand.  For each 32-key tile (staged once in LDS and shared by the 4
// waves) the wave computes the TRANSPOSED score tile S^T = K * Q^T with v_mfma_f32_32x32x2_f32, so a
// lane holds ONE query column (lane&31) and 16 of the 32 keys: the online-softmax statistics are
// per-lane scalars (one __shfl_xor(.,32) per tile to share the row max between the two half-waves),
// and P^T already sits in the B-operand layout of the second product O^T = V^T * P^T - no LDS or
// permute round trip for P.  The key order inside a 2-wide MFMA k-step is the fragment's own row
// order (mfma32_row), which a sum over keys does not care about.
//
// Windowed mode reads q/k/v straight out of the natural (y,x)-ordered token matrix: window
// partition, zero padding (=> padded keys carry k = v = qkv.bias, utils.py:28-32) and unpartition are
// pure index arithmetic here; nothing is materialised.
#include "common.h"

namespace {

constexpr int BQ = 128, BKEYS = 32;

struct RowMap {
  int win, H, W, nwx, L, wins;
  __device__ __forceinline__ long row(int b, int i) const {   // -1 => padded position
    if (win == 0) return (long)b * L + i;
    long base = 0;
    if (wins > 0) { const int img = b / wins; b -= img * wins; base = (long)img * H * W; }
    const int wy = b / nwx, wx = b - wy * nwx;
    const int ly = i / win, lx = i - ly * win;
    const int y = wy * win + ly, x = wx * win + lx;
    return (y < H && x < W) ? base + (long)y * W + x : -1;
  }
};

template <int D, int DV>
__global__ __launch_bounds__(256) void k_attention(AttnArgs a) {
  constexpr int DVP = ((DV + 31) / 32) * 32;
  constexpr int NT = DVP / 32;
  constexpr int KLD = D + 1, VLD = DVP + 1;
  constexpr int NK4 = (8 * D + 255) / 256, NV4 = (8 * DV + 255) / 256;
  __shared__ float Ks[2][BKEYS][KLD];
  __shared__ float Vs[2][BKEYS][VLD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQ;

  const RowMap qm{a.win_q, a.Hq, a.Wq, a.nwx, a.Lq, a.wins};
  const RowMap km{a.win_k, a.Hk, a.Wk, a.nwx, a.Lk, a.wins};

  // ---- stage the block's 128 query rows through LDS into per-lane B-operand registers
  float qreg[D / 2];
  for (int w = 0; w < 4; ++w) {
    for (int idx = tid; idx < 32 * (D / 4); idx += 256) {
      const int r = idx / (D / 4), c4 = idx - r * (D / 4);
      const int qi = q0 + w * 32 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qi < a.Lq) {
        const long row = qm.row(b, qi);
        if (row >= 0) v = *reinterpret_cast<const float4*>(a.q + row * a.ldq + h * D + c4 * 4);
      }
      float* dst = &Ks[0][r][c4 * 4];
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int s = 0; s < D / 2; ++s) qreg[s] = Ks[0][l31][2 * s + half];
    }
    __syncthreads();
  }
  const bool wave_active = (q0 + wave * 32) < a.Lq;

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = a.scale * 1.44269504088896340736f;

  float4 rk[NK4], rv[NV4];
  auto load_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NK4; ++i) {
      const int idx = tid + 256 * i;
      rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < 8 * D) {
        const int r = idx / (D / 4), c4 = idx - r * (D / 4);
        const int ki = kt * BKEYS + r;
        if (ki < a.Lk) {
          const long row = km.row(b, ki);
          const float* p = row >= 0 ? a.k + row * a.ldk + h * D : (a.k_pad ? a.k_pad + h * D : nullptr);
          if (p) rk[i] = *reinterpret_cast<const float4*>(p + c4 * 4);
        }
      }
    }
  };
  auto load_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int idx = tid + 256 * i;
      rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < 8 * DV) {
        const int r = idx / (DV / 4), c4 = idx - r * (DV / 4);
        const int ki = kt * BKEYS + r;
        if (ki < a.Lk) {
          const long row = km.row(b, ki);
          const float* p = row >= 0 ? a.v + row * a.ldv + h * DV : (a.v_pad ? a.v_pad + h * DV : nullptr);
          if (p) rv[i] = *reinterpret_cast<const float4*>(p + c4 * 4);
        }
      }
    }
  };
  auto store_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NK4; ++i) {
      const int idx = tid + 256 * i;
      if (idx < 8 * D) {
        const int r = idx / (D / 4), c4 = idx - r * (D / 4);
        float* dst = &Ks[buf][r][c4 * 4];
        dst[0] = rk[i].x; dst[1] = rk[i].y; dst[2] = rk[i].z; dst[3] = rk[i].w;
      }
    }
  };
  auto store_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int idx = tid + 256 * i;
      if (idx < 8 * DV) {
        const int r = idx / (DV / 4), c4 = idx - r * (DV / 4);
        float* dst = &Vs[buf][r][c4 * 4];
        dst[0] = rv[i].x; dst[1] = rv[i].y; dst[2] = rv[i].z; dst[3] = rv[i].w;
      }
    }
  };

  // zero the DV..DVP padding columns of both V buffers once (they feed wasted MFMA rows only, but
  // must not hold NaNs)
  if (DVP != DV) {
    for (int idx = tid; idx < 2 * BKEYS * (DVP - DV); idx += 256) {
      const int buf = idx / (BKEYS * (DVP - DV)), rem = idx - buf * (BKEYS * (DVP - DV));
      Vs[buf][rem / (DVP - DV)][DV + rem % (DVP - DV)] = 0.f;
    }
  }

  const int nkt = (a.Lk + BKEYS - 1) / BKEYS;
  load_k(0);
  store_k(0);
  load_v(0);
  store_v(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    // prefetch is split in two so K and V share the same staging registers: K(t+1) flies under the
    // S^T MFMAs, V(t+1) under the PV MFMAs.
    if (kt + 1 < nkt) load_k(kt + 1);
    f32x16 acc;
    float alpha = 1.f;
    if (wave_active) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < D / 2; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[cur][l31][2 * s + half], qreg[s], acc, 0, 0, 0);
      // keep the LDS operand reads a few MFMAs ahead instead of letting the scheduler hoist all D/2 of
      // them (that blows the register budget at D=256): 4 reads up front, then 1 read per MFMA.
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int s = 0; s < D / 2; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kt * BKEYS + mfma32_row(e, half);
        acc[e] = key < a.Lk ? acc[e] * sc : -INFINITY;
        tmax = fmaxf(tmax, acc[e]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run, tmax);
      alpha = exp2f(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[e] = exp2f(acc[e] - m_new);
        psum += acc[e];
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
    }
    if (kt + 1 < nkt) {
      store_k(cur ^ 1);
      load_v(kt + 1);
    }
    if (wave_active) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
#pragma unroll
        for (int s = 0; s < 16; ++s)
          o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[cur][mfma32_row(s, half)][t * 32 + l31], acc[s], o[t], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
    if (kt + 1 < nkt) store_v(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if (!wave_active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  const int qi = q0 + wave * 32 + l31;
  if (qi >= a.Lq) return;
  const long orow = qm.row(b, qi);
  if (orow < 0) return;
  float* op = a.o + orow * a.ldo + h * DV;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int dv = t * 32 + mfma32_row(e, half);
      if (dv < DV) op[dv] = o[t][e] * inv;
    }
}

template <int D, int DV>
int launch_t(const AttnArgs& a, hipStream_t st) {
  dim3 grid(cdiv(a.Lq, BQ), a.heads, a.batch);
  hipLaunchKernelGGL((k_attention<D, DV>), grid, dim3(256), 0, st, a);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

}  // namespace

int launch_attention(const AttnArgs& a, hipStream_t st) {
  DS2_REQUIRE(a.batch > 0 && a.heads > 0 && a.Lq > 0 && a.Lk > 0, "attention: bad sizes");
  DS2_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0, "attention: row strides must be multiples of 4");
  DS2_REQUIRE(a.batch <= 65535 && a.heads <= 65535, "attention: grid too large");
  if (attention_fewq_supported(a)) return launch_attention_fewq(a, st);   // few queries x many keys: split-key fp32 path
  if (ds2_split_mode() && attention_winlds_supported(a)) return launch_attention_winlds(a, st);
  if (ds2_split_mode() && attention_smallwin_supported(a)) return launch_attention_smallwin(a, st);
  if (ds2_split_mode()) {
    const int rc = launch_attention_bf16x3(a, st);
    if (rc != DS2_ERR_UNSUPPORTED) return rc;
  }
#define DS2_ATTN_CASE(d, dv) \
  if (a.D == d && a.DV == dv) return launch_t<d, dv>(a, st);
  DS2_ATTN_CASE(256, 256)
  DS2_ATTN_CASE(256, 64)
  DS2_ATTN_CASE(96, 96)
  DS2_ATTN_CASE(72, 72)
  DS2_ATTN_CASE(56, 56)
  DS2_ATTN_CASE(32, 32)
  DS2_ATTN_CASE(16, 16)
#undef DS2_ATTN_CASE
  ds2_set_error("attention: unsupported head dims D=%d DV=%d", a.D, a.DV);
  return DS2_ERR_UNSUPPORTED;
}
