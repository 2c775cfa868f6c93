// Fused two-layer MLP over pre-split bf16x3 operands:  out = (act(X W1^T + b1) W2^T + b2) * gamma + R,  model width D = 256.
//
// Why: with the two GEMMs as separate kernels the hidden activations make a round trip through HBM as two bf16 planes -
// 2 x rows x H x 4 bytes (memory-attention FFN, 65536 rows x 2048: 1.07 GB per layer against 0.2 GB of everything else) -
// and each GEMM pays its own per-tile epilogue (LDS-staged plane split, store burst, the next tile's exposed first DMA):
// at K = 256 the pair is bound by that traffic and those fixed costs, not by the matrix pipe (DESIGN.md "GEMM findings").
// Here the hidden activations never leave the registers.
//
// Formulation (everything transposed so that accumulators ARE operands - the same trick as the attention kernels' P):
//   a workgroup (4 waves, ONE per SIMD, up to 512 VGPRs each) owns 128 token rows; wave w owns tokens [32 w, 32 w + 32).
//   phase A:  Hid^T[h, t] = sum_k W1[h, k] X[t, k]     A operand = W1 rows (LDS), B operand = X rows - held in REGISTERS for
//             the whole row block (16 k-steps x {hi, lo} x 4 VGPRs = 128 VGPRs), accumulator lane = (token, 4 hidden rows)
//   in registers: + b1, activation, split into bf16 hi / lo -> directly the B operand (token, 8 hidden k) of
//   phase B:  Out^T[n, t] += sum_h W2[n, h] Hid[t, h]   A operand = W2 rows (LDS), accumulator 256 n x 32 tokens = 128 VGPRs
//   The accumulator of a 32x32x16 MFMA gives a lane hidden rows {8g + 4h + e}: two g-groups make one 16-deep k-step whose
//   k-slots (half h, s = 0..7) hold hidden units {4h + s | s < 4} u {8 + 4h + (s - 4)} of the 16-group - so W2 is stored with
//   its hidden index permuted the same way inside every group of 16 ([0..3, 8..11, 4..7, 12..15]; done once per weight,
//   launch_mlp256_permute_w2) and its fragments stay plain 16-byte LDS reads.  A sum over k is order-free: exact.
//   The hidden dimension is walked in chunks of 64: per chunk 4 W1 tiles (64 hidden x 64 k, 16 KB with both planes) and
//   2 W2 tiles (256 n x 32 hidden, 32 KB) stream through a 4-slot LDS ring by LDS-DMA (XOR-swizzled source addresses, the
//   GEMM kernels' image), one barrier per tile, counted vmcnt; 24 resp. 48 MFMAs per wave per tile.  Register budget per
//   lane: X 128 + out 128 + hidden accumulators 32 + their bf16 fragments 32 = 320 of 512, the rest buys fragment prefetch.
//   Epilogue: a lane holds 4 consecutive output columns of its token: + b2, * gamma, + R, one 16-byte store - no LDS round trip.
// Product terms and their order as in the GEMM kernels: a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 accumulation.
// X2 (MlpArgs::f16x2, round 4): TWO terms on IEEE fp16 planes - x_h w_l + x_h w_h with x and the hidden activations rounded to one
// fp16 plane (11 bits) and the weights kept as two (22 bits).  The matrix pipe of a fully loaded chip is power-limited
// (profiles/r04_mfma_power_calibration.txt), so a third of the MFMAs is a third of the time; tools/prec_emulate.py f16x2w shows
// no loss against the reference goldens when it is confined to the memory attention / memory encoder (DS2_EMU_ONLY=ma|menc),
// and a visible one in the image encoder - which keeps three terms.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

// 1 (default): the hidden-dimension loop of the ReLU / two-fp16-term instantiations (the memory attention's FFN) is the generated
// inline-assembly statement of tools/gen/gen_mlp256_x4m.py (fragment reads 4 units ahead with counted waits, DMA 6 tiles ahead, one
// barrier per 16 KiB tile, activation as fillers); 0: the C++ loop for every instantiation.  Same arithmetic in the same order per
// element: tools/mlp_layout_check.py compares the two builds bit for bit.
#ifndef DS2_MLP_X4M
#define DS2_MLP_X4M 1
#endif
#if DS2_MLP_X4M
#ifdef X4M_INC_FILE
#include X4M_INC_FILE      // (tools/x4m_variant.sh: ablation / parameter variants of the generated body)
#else
#include "mlp256_x4m_body.inc"
#include "mlp256_x4m_gelu_body.inc"      // the same loop with ds2_gelu's instruction sequence as the activation (memory encoder CXBlock)
#endif
#endif
#ifndef X4M_GELU_BODY                    // (a variant build with X4M_INC_FILE carries the ReLU body only)
#define DS2_MLP_X4M_GELU 0
#else
#ifndef DS2_MLP_X4M_GELU
#define DS2_MLP_X4M_GELU 1
#endif
#endif

// ablation builds for profiling only (tools/ab.py build NAME -DDS2_MLP_ABL=mask; results are WRONG for mask != 0), bits:
// 1 = no bias / activation (split only), 2 = no workgroup barrier per step, 4 = no prefetch DMA inside the loop, 8 = the lo-plane weight
// fragments are not read (half the LDS fragment reads), 16 = no weight fragment reads at all, 32 = the in-loop DMA is issued but never
// waited for (vmcnt), 64 = no main loop (prologue + epilogue only)
#ifndef DS2_MLP_ABL
#define DS2_MLP_ABL 0
#endif
// 1 (default, round 5): the weight tiles sit in LDS as rows of 128 bytes (64 k resp. 64 hidden units) and a DMA piece is 8 rows x 128 B -
// every piece requests FULL L2 lines.  0: rows of 64 bytes, pieces of 16 rows x 64 B = half lines (rounds 3 - 4).  The L2 -> LDS path is bound
// by line requests (DESIGN.md "GEMM, round 5"); the ablation without the in-loop DMA runs 22 % faster.  Same MFMA order per element: bit-identical.
#ifndef DS2_MLP_LINE128
#define DS2_MLP_LINE128 1
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {   // v_cvt_pk_f16_f32, round to nearest even
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2{ds2_sat_f16(a), ds2_sat_f16(b)}), h2));   // (saturating)
}
constexpr int MD = 256;            // model width (k of phase A, n of phase B)
constexpr int MBR = 128;           // token rows per workgroup
constexpr int MHC = 64;            // hidden units per chunk
constexpr int MSLOT = 32768;       // one ring slot: hi plane at +0, lo plane at +16384; rows of 64 bytes (32 k)
constexpr int MLO = 16384;
constexpr int MNS = 4;             // ring slots: three tiles ahead of the one being computed are in flight
constexpr int MTA = MD / 64;       // W1 tiles per chunk: 64 hidden x 64 k (two 32-k sub-tiles of 4 KB per plane)
constexpr int MTB = MHC / 32;      // W2 tiles per chunk: 256 n x 32 hidden
constexpr int MT_PER_CHUNK = MTA + MTB;   // 6
constexpr int MT_PAIR = 2 * MT_PER_CHUNK; // the loop body covers a PAIR of chunks: 12 tiles, a multiple of the ring size
static_assert(MT_PAIR % MNS == 0, "the ring slot of a tile must depend on its position in the chunk pair only");

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// hidden index permutation inside a group of 16 (see the header): new position p holds old index perm16(p)
__host__ __device__ inline int perm16(int p) { return (p & 3) | ((p & 4) << 1) | ((p & 8) >> 1); }

__global__ void k_mlp256_permute_w2(const float* w2, int ldw, int n_rows, int H, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_rows * H) return;
  const int c = (int)(i % H), r = (int)(i / H);
  out[(size_t)r * H + c] = w2[(size_t)r * ldw + (c & ~15) + perm16(c & 15)];
}

// DMA instructions one wave issues for a tile (both planes): W1 tile 16 pieces / 4 waves, W2 tile 32 pieces / 4 waves
__host__ __device__ constexpr int tile_dma(int pos) { return (pos % MT_PER_CHUNK) < MTA ? 4 : 8; }

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {   // s_waitcnt needs an immediate
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
  else static_assert(N == 8 || N == 12 || N == 16, "unexpected DMA count");
}

// LNF: the LayerNorm of the result rows in the epilogue is compiled in (MlpArgs::ln_w).  A template parameter since round 5: with the
// LayerNorm code present EVERY instantiation - all at the full 512 registers - spilled 29 - 40 VGPRs to scratch;
// without it none does, so only the one instantiation that uses the fusion (ReLU, two fp16 terms: the memory attention's FFN) carries it.
template <int ACT, bool X2, bool LNF, bool LNI = false>
__global__ __launch_bounds__(256, 1) void k_mlp256(MlpArgs a) {
  using FragT = typename std::conditional<X2, f16x8, bf16x8>::type;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  float* b1s = reinterpret_cast<float*>(lds + MNS * MSLOT);   // [H]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  typedef __attribute__((address_space(3))) void* lds_ptr;

  for (int i = tid; i < a.H; i += 256) b1s[i] = a.b1 ? a.b1[i] : 0.f;
  // LayerNorm weight / bias of the fused epilogue: staged in LDS, because read from global memory the compiler hoists the 64 per-lane
  // 64-bit addresses (ln_w + n, ln_b + n for the 32 column groups) out of the row-block loop and spills them (29 - 40 VGPRs of scratch)
  float* lns = b1s + a.H;   // [4][MD]: ln_w, ln_b, b2, gamma (the same holds for b2 / gamma of the plain epilogue)
  if (LNF && a.ln_w) {
    lns[tid] = a.ln_w[tid];
    lns[MD + tid] = a.ln_b[tid];
  }
  lns[2 * MD + tid] = a.b2 ? a.b2[tid] : 0.f;
  lns[3 * MD + tid] = a.gamma ? a.gamma[tid] : 1.f;
  if constexpr (LNI) {      // weight / bias of the input LayerNorm: [4 MD, 6 MD)
    lns[4 * MD + tid] = a.lni_w[tid];
    lns[5 * MD + tid] = a.lni_b[tid];
  }

  const char* w1h = reinterpret_cast<const char*>(a.W1_hi);
  const char* w1l = reinterpret_cast<const char*>(a.W1_lo);
  const char* w2h = reinterpret_cast<const char*>(a.W2_hi);
  const char* w2l = reinterpret_cast<const char*>(a.W2_lo);
  // (hidden split: this workgroup walks the chunks [c_begin, c_end) of its part)
  const int nchunk_all = a.H / MHC, nparts = a.hsplit > 1 ? a.hsplit : 1;
  const int hp = nparts > 1 ? (int)blockIdx.y : 0;
  const int c_begin = hp * (nchunk_all / nparts), c_end = c_begin + nchunk_all / nparts;
  // DMA: a piece = 16 rows x 64 B of one plane (1 KiB, one wave instruction); lane -> (row = lane >> 2, physical 16-byte
  // chunk = lane & 3) fetching the logical chunk (lane & 3) ^ ((lane >> 4) & 3)
#if DS2_MLP_LINE128
  // DMA: a piece = 8 rows x 128 B of one plane (1 KiB, one wave instruction); lane -> (row = lane >> 3, physical 16-byte chunk = lane & 7)
  // fetching the logical chunk (lane & 7) ^ ((tile row >> 1) & 7); the reader applies the same XOR
  const int drow = lane >> 3;
  const int sw = (l31 >> 1) & 7;
  unsigned offA[2], offB[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {   // W1 piece pc = 2 wave + j: tile rows 8 pc .. 8 pc + 7 (hidden units), 64 k per row
    const int row = (wave * 2 + j) * 8 + drow;
    offA[j] = ((unsigned)row * (unsigned)a.ldw1 + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2u;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {   // W2 piece pc = 4 wave + j: tile rows 8 pc .. (output columns n of this half), 64 hidden units per row
    const int row = (wave * 4 + j) * 8 + drow;
    offB[j] = ((unsigned)row * (unsigned)a.ldw2 + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2u;
  }
  const unsigned strideB = 128u * (unsigned)a.ldw2 * 2u;            // bytes between the two n-halves of W2
#else
  const int drow = lane >> 2;
  const int dlc = (lane & 3) ^ ((lane >> 4) & 3);
  const int sw = (l31 >> 2) & 3;   // reader side of the same involution
  // per-lane source offsets (bytes) of this wave's pieces, chunk 0 / tile 0
  unsigned offA[2], offB[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {   // W1 piece pc = 2 wave + j: sub-tile u = pc >> 2 (k + 32 u), rows (pc & 3) * 16 ..
    const int pc = wave * 2 + j;
    offA[j] = ((unsigned)((pc & 3) * 16 + drow) * (unsigned)a.ldw1 + (pc >> 2) * 32 + dlc * 8) * 2u;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = ((unsigned)((wave * 4 + j) * 16 + drow) * (unsigned)a.ldw2 + dlc * 8) * 2u;
  const unsigned strideB = 0;
#endif
  const unsigned strideA = (unsigned)MHC * (unsigned)a.ldw1 * 2u;   // bytes between chunks of W1 (64 hidden rows)

  // tile at position P6 (0..5) of chunk `c` into the ring slot of pair position PP (slot = PP % MNS)
#define MLP_DMA(P6, c, PP)                                                                                                \
  {                                                                                                                       \
    unsigned char* base_ = lds + ((PP) % MNS) * MSLOT;                                                                    \
    if constexpr ((P6) < MTA) {                                                                                           \
      const unsigned o_ = (unsigned)(c) * strideA + (P6) * 128u;                                                          \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                     \
        __builtin_amdgcn_global_load_lds(w1h + (offA[j] + o_), (lds_ptr)(base_ + (wave * 2 + j) * 1024), 16, 0, 0);       \
        __builtin_amdgcn_global_load_lds(w1l + (offA[j] + o_), (lds_ptr)(base_ + MLO + (wave * 2 + j) * 1024), 16, 0, 0); \
      }                                                                                                                   \
    } else {                                                                                                              \
      const unsigned o_ = DS2_MLP_LINE128 ? (unsigned)(c) * MHC * 2u + ((P6) - MTA) * strideB                             \
                                          : ((unsigned)(c) * MHC + ((P6) - MTA) * 32u) * 2u;                              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                     \
        __builtin_amdgcn_global_load_lds(w2h + (offB[j] + o_), (lds_ptr)(base_ + (wave * 4 + j) * 1024), 16, 0, 0);       \
        __builtin_amdgcn_global_load_lds(w2l + (offB[j] + o_), (lds_ptr)(base_ + MLO + (wave * 4 + j) * 1024), 16, 0, 0); \
      }                                                                                                                   \
    }                                                                                                                     \
  }
  // prefetch issued at pair position PP: the tile MNS - 1 positions ahead (in this pair, or in the next one)
#define MLP_PREFETCH(PP)                                                                                                  \
  if constexpr ((DS2_MLP_ABL & 4) != 0) {} else if constexpr ((PP) + MNS - 1 < MT_PAIR) MLP_DMA(((PP) + MNS - 1) % MT_PER_CHUNK, c2 + ((PP) + MNS - 1) / MT_PER_CHUNK, (PP) + MNS - 1) \
  else MLP_DMA(((PP) + MNS - 1) % MT_PER_CHUNK, cn2 + ((PP) + MNS - 1 - MT_PAIR) / MT_PER_CHUNK, (PP) + MNS - 1)
  // order of a step's instructions (one wave per SIMD: nobody else hides the LDS latency): the DMA issue first, then the
  // fragment pairs (hi, lo) in groups of two = 6 MFMAs.  Three register sets: groups g and g + 1 are resident, group g + 2
  // is fetched into the third set after the first two MFMAs of group g - hipcc's wait before group g + 1 is an
  // lgkmcnt(0), so the youngest read in flight must be old by then (4 MFMAs = 128 cycles).  NP = pairs of the step.
#define MLP_INTERLEAVE(NP)                                                                    \
  __builtin_amdgcn_sched_group_barrier(0x010, 16, 0);   /* VMEM: the prefetch DMA */           \
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    /* DS read: groups 0, 1 */             \
  _Pragma("unroll") for (int i_ = 0; i_ < (NP) / 2 - 2; ++i_) {                               \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, X2 ? 2 : 4, 0);                                \
  }                                                                                           \
  __builtin_amdgcn_sched_group_barrier(0x008, X2 ? 8 : 12, 0);
  // end of the step at pair position PP: the next tile must have landed (the two after it may stay in flight), this
  // wave's fragment reads are retired, then everybody meets
#define MLP_STEP_END(PP)                                                         \
  __builtin_amdgcn_sched_barrier(0);                                             \
  if constexpr ((DS2_MLP_ABL & 36) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* 6: DMA issued, never waited for */ \
  else wait_vm_lgkm<tile_dma((PP) + 2) + tile_dma((PP) + 3)>();                  \
  if constexpr ((DS2_MLP_ABL & 2) == 0) __builtin_amdgcn_s_barrier();                  \
  __builtin_amdgcn_sched_barrier(0);

#if DS2_MLP_X4M
  constexpr bool X4M = (ACT == DS2_ACT_RELU || (ACT == DS2_ACT_GELU && DS2_MLP_X4M_GELU)) && X2 && DS2_MLP_ABL == 0;
#else
  constexpr bool X4M = false;
#endif
  // operands of the assembly loop: tiles of 64 rows x 128 B (a wave issues pieces 2 wave, 2 wave + 1 of both planes), fragment reads at
  // row l31, 16-byte chunk (2 k + half) ^ ((l31 >> 1) & 7) - the k-step enters as an XOR of 32 k on the byte address
  unsigned m_rd0 = 0, m_offa[2] = {0, 0}, m_offb[2] = {0, 0}, m_baddr = 0, m_ldsb = 0, m_stridea = 0, m_strideb = 0, m_nch = 0;
  unsigned long long m_w1h = 0, m_w1l = 0, m_w2h = 0, m_w2l = 0;
  if constexpr (X4M) {
    m_ldsb = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)lds));
    m_rd0 = m_ldsb + (unsigned)l31 * 128u + (unsigned)((half ^ ((l31 >> 1) & 7)) << 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 8 + (lane >> 3);
      const unsigned ch = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 8);
      m_offa[j] = ((unsigned)row * (unsigned)a.ldw1 + ch) * 2u;
      m_offb[j] = ((unsigned)row * (unsigned)a.ldw2 + ch) * 2u;
    }
    m_baddr = m_ldsb + (unsigned)(MNS * MSLOT) + (unsigned)(c_begin * MHC + 4 * half) * 4u;
    m_stridea = (unsigned)MHC * (unsigned)a.ldw1 * 2u;
    m_strideb = 64u * (unsigned)a.ldw2 * 2u;
    m_nch = (unsigned)(c_end - c_begin);
    m_w1h = reinterpret_cast<unsigned long long>(w1h) + (unsigned long long)c_begin * m_stridea;
    m_w1l = reinterpret_cast<unsigned long long>(w1l) + (unsigned long long)c_begin * m_stridea;
    m_w2h = reinterpret_cast<unsigned long long>(w2h) + (unsigned long long)c_begin * (MHC * 2u);
    m_w2l = reinterpret_cast<unsigned long long>(w2l) + (unsigned long long)c_begin * (MHC * 2u);
  }
  const int nrb = (a.rows + MBR - 1) / MBR;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int tok = rb * MBR + wave * 32 + l31;
    const int tokc = tok < a.rows ? tok : a.rows - 1;   // clamp: rows beyond the end are computed but never stored
    // ---- X fragments of this wave's 32 tokens: B operand (token = lane & 31, k = 16 s + 8 half .. + 7), both planes
    FragT xh[MD / 16], xl[X2 ? 1 : MD / 16];
    if constexpr (LNI) {
      static_assert(!LNI || X2, "the input LayerNorm is built for the two-fp16-term form");
      if (rb == (int)blockIdx.x) __syncthreads();   // (the staged LayerNorm weight / bias: first row block only)
      // this lane's half of the token's row: columns 16 s + 8 half + j, j = 0..7 - the float4 groups 4 s + 2 half + jj of k_layernorm_vec
      const float* px = a.X_f32 + (size_t)tokc * a.ldxf + half * 8;
      float4 v[MD / 16][2];
#pragma unroll
      for (int s = 0; s < MD / 16; ++s) {
        v[s][0] = *reinterpret_cast<const float4*>(px + s * 16);
        v[s][1] = *reinterpret_cast<const float4*>(px + s * 16 + 4);
      }
      // wave_sum's butterfly (xor 32, 16, 8, 4, 2, 1 over the 64 float4 groups of the row) as a tree: group index L = 4 s + 2 half + jj, so
      // xor 32 / 16 / 8 / 4 pair s with s ^ 8 / 4 / 2 / 1 (in this lane), xor 2 is the other half (lane ^ 32), xor 1 is jj
      auto both = [](float x) {          // x + (the same quantity of lane ^ 32): the pair holds both, addition commutes
        const unsigned u = __float_as_uint(x);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
      };
      auto tree = [&](float (&p)[MD / 16][2]) {
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1)
#pragma unroll
          for (int s = 0; s < d; ++s) {
            p[s][0] = p[s][0] + p[s + d][0];
            p[s][1] = p[s][1] + p[s + d][1];
          }
        const float t0 = both(p[0][0]), t1 = both(p[0][1]);
        return t0 + t1;
      };
      float ps[MD / 16][2];
#pragma unroll
      for (int s = 0; s < MD / 16; ++s)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) ps[s][jj] = (v[s][jj].x + v[s][jj].y) + (v[s][jj].z + v[s][jj].w);
      const float mean = tree(ps) / (float)MD;
#pragma unroll
      for (int s = 0; s < MD / 16; ++s)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const float d0 = v[s][jj].x - mean, d1 = v[s][jj].y - mean, d2 = v[s][jj].z - mean, d3 = v[s][jj].w - mean;
          ps[s][jj] = 0.f + (__builtin_fmaf(d0, d0, d1 * d1) + __builtin_fmaf(d2, d2, d3 * d3));   // (k_layernorm_vec's contraction, spelled out)
        }
      const float rstd = 1.f / sqrtf(tree(ps) / (float)MD + a.lni_eps);
#pragma unroll
      for (int s = 0; s < MD / 16; ++s) {
        unsigned r[4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int c = 16 * s + 8 * half + 4 * jj;
          const float4 w4 = *reinterpret_cast<const float4*>(lns + 4 * MD + c), b4 = *reinterpret_cast<const float4*>(lns + 5 * MD + c);
          float4 o;
          o.x = (v[s][jj].x - mean) * rstd * w4.x + b4.x;
          o.y = (v[s][jj].y - mean) * rstd * w4.y + b4.y;
          o.z = (v[s][jj].z - mean) * rstd * w4.z + b4.z;
          o.w = (v[s][jj].w - mean) * rstd * w4.w + b4.w;
          // the planes a LayerNorm pass would store (k_layernorm_vec), then their sum rounded to the one fp16 plane as below
          const unsigned h0 = cvt_pk_bf16(o.x, o.y), h1 = cvt_pk_bf16(o.z, o.w);
          const unsigned l0 = cvt_pk_bf16(o.x - bf_lo(h0), o.y - bf_hi(h0)), l1 = cvt_pk_bf16(o.z - bf_lo(h1), o.w - bf_hi(h1));
          r[2 * jj] = cvt_pk_f16(bf_lo(h0) + bf_lo(l0), bf_hi(h0) + bf_hi(l0));
          r[2 * jj + 1] = cvt_pk_f16(bf_lo(h1) + bf_lo(l1), bf_hi(h1) + bf_hi(l1));
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        xh[s] = __builtin_bit_cast(FragT, (u32x4{r[0], r[1], r[2], r[3]}));
      }
    } else {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4* ph = reinterpret_cast<const u32x4*>(a.X_hi + (size_t)tokc * a.ldx + half * 8);
      const u32x4* pl = reinterpret_cast<const u32x4*>(a.X_lo + (size_t)tokc * a.ldx + half * 8);
#pragma unroll
      for (int s = 0; s < MD / 16; ++s) {
        const u32x4 h = __builtin_nontemporal_load(ph + s * 2);   // streamed once: keep the weights in L2
        const u32x4 l = __builtin_nontemporal_load(pl + s * 2);
        if constexpr (X2) {   // x = hi + lo of its bf16 planes, rounded to ONE fp16 plane
          u32x4 r;
#pragma unroll
          for (int w = 0; w < 4; ++w) r[w] = cvt_pk_f16(bf_lo(h[w]) + bf_lo(l[w]), bf_hi(h[w]) + bf_hi(l[w]));
          xh[s] = __builtin_bit_cast(FragT, r);
        } else {
          xh[s] = __builtin_bit_cast(FragT, h);
          xl[s] = __builtin_bit_cast(FragT, l);
        }
      }
    }
    f32x16 out[MD / 32];
#pragma unroll
    for (int i = 0; i < MD / 32; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) out[i][e] = 0.f;

#if DS2_MLP_X4M
    if constexpr (X4M) {
      // the whole hidden loop of this row block, ring prologue included (mlp256_x4m_body.inc); every wave leaves it with its DMA still
      // in flight (the wrapped-around tail) - drained below like the C++ loop's
#if DS2_MLP_X4M_GELU
      if constexpr (ACT == DS2_ACT_GELU) {
        asm volatile(X4M_GELU_BODY
                     : [o0] "+a"(out[0]), [o1] "+a"(out[1]), [o2] "+a"(out[2]), [o3] "+a"(out[3]), [o4] "+a"(out[4]), [o5] "+a"(out[5]),
                       [o6] "+a"(out[6]), [o7] "+a"(out[7])
                     : [x0] "v"(xh[0]), [x1] "v"(xh[1]), [x2] "v"(xh[2]), [x3] "v"(xh[3]), [x4] "v"(xh[4]), [x5] "v"(xh[5]), [x6] "v"(xh[6]),
                       [x7] "v"(xh[7]), [x8] "v"(xh[8]), [x9] "v"(xh[9]), [x10] "v"(xh[10]), [x11] "v"(xh[11]), [x12] "v"(xh[12]),
                       [x13] "v"(xh[13]), [x14] "v"(xh[14]), [x15] "v"(xh[15]), [rd0] "v"(m_rd0), [offa0] "v"(m_offa[0]), [offa1] "v"(m_offa[1]),
                       [offb0] "v"(m_offb[0]), [offb1] "v"(m_offb[1]), [baddr] "v"(m_baddr), [w1h] "s"(m_w1h), [w1l] "s"(m_w1l), [w2h] "s"(m_w2h),
                       [w2l] "s"(m_w2l), [stridea] "s"(m_stridea), [strideb] "s"(m_strideb), [ldsb] "s"(m_ldsb), [wave] "s"(wave), [nch] "s"(m_nch)
                     : X4M_GELU_CLOBBERS);
      } else
#endif
      {
        asm volatile(X4M_BODY
                     : [o0] "+a"(out[0]), [o1] "+a"(out[1]), [o2] "+a"(out[2]), [o3] "+a"(out[3]), [o4] "+a"(out[4]), [o5] "+a"(out[5]),
                       [o6] "+a"(out[6]), [o7] "+a"(out[7])
                     : [x0] "v"(xh[0]), [x1] "v"(xh[1]), [x2] "v"(xh[2]), [x3] "v"(xh[3]), [x4] "v"(xh[4]), [x5] "v"(xh[5]), [x6] "v"(xh[6]),
                       [x7] "v"(xh[7]), [x8] "v"(xh[8]), [x9] "v"(xh[9]), [x10] "v"(xh[10]), [x11] "v"(xh[11]), [x12] "v"(xh[12]),
                       [x13] "v"(xh[13]), [x14] "v"(xh[14]), [x15] "v"(xh[15]), [rd0] "v"(m_rd0), [offa0] "v"(m_offa[0]), [offa1] "v"(m_offa[1]),
                       [offb0] "v"(m_offb[0]), [offb1] "v"(m_offb[1]), [baddr] "v"(m_baddr), [w1h] "s"(m_w1h), [w1l] "s"(m_w1l), [w2h] "s"(m_w2h),
                       [w2l] "s"(m_w2l), [stridea] "s"(m_stridea), [strideb] "s"(m_strideb), [ldsb] "s"(m_ldsb), [wave] "s"(wave), [nch] "s"(m_nch)
                     : X4M_CLOBBERS);
      }
    } else
#endif
    {
    // ---- ring prologue: tiles 0..2 of chunk 0 (the X loads above are ordinary loads: drained with them)
    MLP_DMA(0, c_begin, 0)
    MLP_DMA(1, c_begin, 1)
    MLP_DMA(2, c_begin, 2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // (also publishes b1s on the first row block)
    __builtin_amdgcn_sched_barrier(0);

    for (int c2 = c_begin; c2 < ((DS2_MLP_ABL & 64) ? c_begin : c_end); c2 += 2) {
      const int cn2 = (c2 + 2 < c_end) ? c2 + 2 : c_begin;   // (the tail prefetches wrap around: uniform DMA accounting)
      // phase A step at pair position PP (tile T = PP % 6 of chunk c2 + PP / 6): W1 tile, 64 hidden x 64 k
#define MLP_A(PP)                                                                                                         \
  {                                                                                                                       \
    MLP_PREFETCH(PP)                                                                                                      \
    const unsigned char* base_ = lds + ((PP) % MNS) * MSLOT;                                                              \
    FragT wh_[4][MHC / 32], wl_[4][MHC / 32];                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                      \
      _Pragma("unroll") for (int hb = 0; hb < MHC / 32; ++hb) {                                                           \
        const unsigned char* r_ = DS2_MLP_LINE128 ? base_ + (hb * 32 + l31) * 128 + (((ks * 2 + half) ^ sw) << 4)        \
            : base_ + (ks >> 1) * 4096 + (hb * 32 + l31) * 64 + ((((ks & 1) * 2 + half) ^ sw) << 4);                      \
        if constexpr ((DS2_MLP_ABL & 16) != 0) wh_[ks][hb] = xh[ks]; else wh_[ks][hb] = *reinterpret_cast<const FragT*>(r_);     \
        if constexpr ((DS2_MLP_ABL & 24) != 0) wl_[ks][hb] = wh_[ks][hb]; else wl_[ks][hb] = *reinterpret_cast<const FragT*>(r_ + MLO); \
      }                                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                      \
      _Pragma("unroll") for (int hb = 0; hb < MHC / 32; ++hb) {                                                           \
        hid[hb] = mfma16(wl_[ks][hb], xh[((PP) % MT_PER_CHUNK) * 4 + ks], hid[hb]);                                       \
        if constexpr (!X2) hid[hb] = mfma16(wh_[ks][hb], xl[((PP) % MT_PER_CHUNK) * 4 + ks], hid[hb]);                    \
        hid[hb] = mfma16(wh_[ks][hb], xh[((PP) % MT_PER_CHUNK) * 4 + ks], hid[hb]);                                       \
      }                                                                                                                   \
    MLP_INTERLEAVE(8)                                                                                                     \
    MLP_STEP_END(PP)                                                                                                      \
  }
      // phase B step at pair position PP.  DS2_MLP_LINE128: W2 tile = output columns [128 Q, 128 Q + 128) x the chunk's 64 (permuted) hidden
      // units, Q = PP % 6 - 4; its 16 fragment pairs are (k-step t = 0..3, column block nb = 0..3), t outermost - an output element
      // still sums its hidden units in ascending order.  Otherwise: 256 n x 32 hidden units (hidden block Q), pairs (t = 0..1, nb = 0..7).
#if DS2_MLP_LINE128
#define MLP_B(PP)                                                                                                         \
  {                                                                                                                       \
    MLP_PREFETCH(PP)                                                                                                      \
    const unsigned char* base_ = lds + ((PP) % MNS) * MSLOT;                                                              \
    constexpr int Q_ = (PP) % MT_PER_CHUNK - MTA;                                                                         \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                                                    \
      FragT wh_[8], wl_[8];                                                                                               \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                     \
        const int t = t2 * 2 + (i >> 2), nb = i & 3;                                                                      \
        const unsigned char* r_ = base_ + (nb * 32 + l31) * 128 + (((t * 2 + half) ^ sw) << 4);                           \
        if constexpr ((DS2_MLP_ABL & 16) != 0) wh_[i] = xh[i]; else wh_[i] = *reinterpret_cast<const FragT*>(r_);                \
        if constexpr ((DS2_MLP_ABL & 24) != 0) wl_[i] = wh_[i]; else wl_[i] = *reinterpret_cast<const FragT*>(r_ + MLO);         \
      }                                                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                     \
        const int t = t2 * 2 + (i >> 2), nb = Q_ * 4 + (i & 3);                                                           \
        out[nb] = mfma16(wl_[i], fh[t >> 1][t & 1], out[nb]);                                                             \
        if constexpr (!X2) out[nb] = mfma16(wh_[i], fl[t >> 1][t & 1], out[nb]);                                          \
        out[nb] = mfma16(wh_[i], fh[t >> 1][t & 1], out[nb]);                                                             \
      }                                                                                                                   \
    }                                                                                                                     \
    MLP_INTERLEAVE(16)                                                                                                    \
    MLP_STEP_END(PP)                                                                                                      \
  }
#else
#define MLP_B(PP)                                                                                                         \
  {                                                                                                                       \
    MLP_PREFETCH(PP)                                                                                                      \
    const unsigned char* base_ = lds + ((PP) % MNS) * MSLOT;                                                              \
    constexpr int Q_ = (PP) % MT_PER_CHUNK - MTA;                                                                         \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                       \
      FragT wh_[MD / 32], wl_[MD / 32];                                                                                   \
      _Pragma("unroll") for (int nb = 0; nb < MD / 32; ++nb) {                                                            \
        const unsigned char* r_ = base_ + (nb * 32 + l31) * 64 + (((t * 2 + half) ^ sw) << 4);                            \
        if constexpr ((DS2_MLP_ABL & 16) != 0) wh_[nb] = xh[nb]; else wh_[nb] = *reinterpret_cast<const FragT*>(r_);             \
        if constexpr ((DS2_MLP_ABL & 24) != 0) wl_[nb] = wh_[nb]; else wl_[nb] = *reinterpret_cast<const FragT*>(r_ + MLO);      \
      }                                                                                                                   \
      _Pragma("unroll") for (int nb = 0; nb < MD / 32; ++nb) {                                                            \
        out[nb] = mfma16(wl_[nb], fh[Q_][t], out[nb]);                                                                    \
        if constexpr (!X2) out[nb] = mfma16(wh_[nb], fl[Q_][t], out[nb]);                                                 \
        out[nb] = mfma16(wh_[nb], fh[Q_][t], out[nb]);                                                                    \
      }                                                                                                                   \
    }                                                                                                                     \
    MLP_INTERLEAVE(16)                                                                                                    \
    MLP_STEP_END(PP)                                                                                                      \
  }
#endif
      // bias + activation + split of chunk c: hid[hb] registers 8t .. 8t+7 (g = 2t, 2t+1) -> k-step t of hidden block hb.
      // b1 comes from LDS by inline asm: a plain LDS read here makes hipcc drain the DMA ring (s_waitcnt vmcnt(0)).
#define MLP_ACT(c)                                                                                                        \
  {                                                                                                                       \
    f32x4 bb[MHC / 32][4];                                                                                                \
    const unsigned baddr = (unsigned)(MNS * MSLOT) + (unsigned)((c) * MHC + 4 * half) * 4u;                               \
    _Pragma("unroll") for (int hb = 0; hb < MHC / 32; ++hb)                                                               \
      _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bb[hb][g]) : "v"(baddr), "n"((hb * 32 + 8 * g) * 4));         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    _Pragma("unroll") for (int hb = 0; hb < MHC / 32; ++hb)                                                               \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                     \
        float v[8];                                                                                                       \
        _Pragma("unroll") for (int gg = 0; gg < 2; ++gg)                                                                  \
          _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                   \
            v[gg * 4 + e] = (DS2_MLP_ABL & 1) ? hid[hb][(2 * t + gg) * 4 + e]                                              \
                                             : ds2_act(hid[hb][(2 * t + gg) * 4 + e] + bb[hb][2 * t + gg][e], ACT);       \
        uint4 h, l;                                                                                                       \
        if constexpr (X2) {                                                                                               \
          h.x = cvt_pk_f16(v[0], v[1]); h.y = cvt_pk_f16(v[2], v[3]);                                                     \
          h.z = cvt_pk_f16(v[4], v[5]); h.w = cvt_pk_f16(v[6], v[7]);                                                     \
          fh[hb][t] = __builtin_bit_cast(FragT, h);                                                                       \
        } else {                                                                                                          \
          h.x = cvt_pk_bf16(v[0], v[1]); h.y = cvt_pk_bf16(v[2], v[3]);                                                   \
          h.z = cvt_pk_bf16(v[4], v[5]); h.w = cvt_pk_bf16(v[6], v[7]);                                                   \
          l.x = cvt_pk_bf16(v[0] - bf_lo(h.x), v[1] - bf_hi(h.x));                                                        \
          l.y = cvt_pk_bf16(v[2] - bf_lo(h.y), v[3] - bf_hi(h.y));                                                        \
          l.z = cvt_pk_bf16(v[4] - bf_lo(h.z), v[5] - bf_hi(h.z));                                                        \
          l.w = cvt_pk_bf16(v[6] - bf_lo(h.w), v[7] - bf_hi(h.w));                                                        \
          fh[hb][t] = __builtin_bit_cast(FragT, h);                                                                       \
          fl[hb][t] = __builtin_bit_cast(FragT, l);                                                                       \
        }                                                                                                                 \
      }                                                                                                                   \
  }
#define MLP_ZERO_HID()                                                   \
  _Pragma("unroll") for (int i = 0; i < MHC / 32; ++i)                   \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) hid[i][e] = 0.f;
      static_assert(MTA == 4 && MTB == 2, "the chunk pair is unrolled by hand");
      f32x16 hid[MHC / 32];
      FragT fh[MHC / 32][2], fl[X2 ? 1 : MHC / 32][2];
      MLP_ZERO_HID()
      MLP_A(0) MLP_A(1) MLP_A(2) MLP_A(3)
      MLP_ACT(c2)
      MLP_B(4) MLP_B(5)
      MLP_ZERO_HID()
      MLP_A(6) MLP_A(7) MLP_A(8) MLP_A(9)
      MLP_ACT(c2 + 1)
      MLP_B(10) MLP_B(11)
    }
    }
    // ---- the ring still holds the (wrapped-around) prefetches: drain them before the ordinary loads / stores of the
    //      epilogue and before the next row block restarts the ring
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: lane (token, half) holds out[token][nb*32 + 8g + 4 half + e], e = 0..3: one float4 per (nb, g)
    if (!LNF && nparts > 1) {   // hidden split: the raw partial sums of this part; k_mlp256_merge finishes (never with LNF: launch_mlp256)
      if (tok < a.rows) {
        float* pp = a.part + ((size_t)hp * a.rows + tok) * MD;
#pragma unroll
        for (int nb = 0; nb < MD / 32; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(pp + nb * 32 + 8 * g + 4 * half) =
                make_float4(out[nb][g * 4 + 0], out[nb][g * 4 + 1], out[nb][g * 4 + 2], out[nb][g * 4 + 3]);
      }
      continue;
    }
    const bool ln = LNF && a.ln_w != nullptr;
    float rsum = 0.f;
#pragma unroll
    for (int nb = 0; nb < MD / 32; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb * 32 + 8 * g + 4 * half;
        float4 v = make_float4(out[nb][g * 4 + 0], out[nb][g * 4 + 1], out[nb][g * 4 + 2], out[nb][g * 4 + 3]);
        if (a.b2) {
          const float4 b = *reinterpret_cast<const float4*>(lns + 2 * MD + n);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (a.gamma) {
          const float4 gm = *reinterpret_cast<const float4*>(lns + 3 * MD + n);
          v.x *= gm.x; v.y *= gm.y; v.z *= gm.z; v.w *= gm.w;
        }
        if (a.R && tok < a.rows) {
          const f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.R + (size_t)tok * a.ldr + n));
          v.x += r[0]; v.y += r[1]; v.z += r[2]; v.w += r[3];
        }
        if (tok < a.rows) *reinterpret_cast<float4*>(a.out + (size_t)tok * a.ldo + n) = v;
        if (ln) {   // keep the final values for the statistics (the accumulators are dead otherwise)
          out[nb][g * 4 + 0] = v.x; out[nb][g * 4 + 1] = v.y; out[nb][g * 4 + 2] = v.z; out[nb][g * 4 + 3] = v.w;
          rsum += (v.x + v.y) + (v.z + v.w);
        } else if (a.out_hi && tok < a.rows) {   // (same split as k_split_rows: round to nearest even, lo = the rounded remainder)
          uint2 h, l;
          h.x = cvt_pk_bf16(v.x, v.y);
          h.y = cvt_pk_bf16(v.z, v.w);
          l.x = cvt_pk_bf16(v.x - __uint_as_float(h.x << 16), v.y - __uint_as_float(h.x & 0xffff0000u));
          l.y = cvt_pk_bf16(v.z - __uint_as_float(h.y << 16), v.w - __uint_as_float(h.y & 0xffff0000u));
          *reinterpret_cast<uint2*>(a.out_hi + (size_t)tok * a.ldop + n) = h;
          *reinterpret_cast<uint2*>(a.out_lo + (size_t)tok * a.ldop + n) = l;
        }
      }
    }
    if (ln) {   // LayerNorm over the token's 256 values: 128 in this lane, 128 in lane ^ 32; two-pass statistics as k_layernorm_vec
      auto both = [](float x) {
        const unsigned u = __float_as_uint(x);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
      };
      const float mean = both(rsum) * (1.f / MD);
      float q = 0.f;
#pragma unroll
      for (int nb = 0; nb < MD / 32; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float d = out[nb][e] - mean;
          q += d * d;
        }
      const float rstd = 1.f / sqrtf(both(q) * (1.f / MD) + a.ln_eps);
      if (tok < a.rows) {
#pragma unroll
        for (int nb = 0; nb < MD / 32; ++nb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nb * 32 + 8 * g + 4 * half;
            const float4 w4 = *reinterpret_cast<const float4*>(lns + n), b4 = *reinterpret_cast<const float4*>(lns + MD + n);
            float4 y;
            y.x = (out[nb][g * 4 + 0] - mean) * rstd * w4.x + b4.x;
            y.y = (out[nb][g * 4 + 1] - mean) * rstd * w4.y + b4.y;
            y.z = (out[nb][g * 4 + 2] - mean) * rstd * w4.z + b4.z;
            y.w = (out[nb][g * 4 + 3] - mean) * rstd * w4.w + b4.w;
            if (a.ln_out) *reinterpret_cast<float4*>(a.ln_out + (size_t)tok * a.ldln + n) = y;
            if (a.out_hi) {
              uint2 h, l;
              h.x = cvt_pk_bf16(y.x, y.y);
              h.y = cvt_pk_bf16(y.z, y.w);
              l.x = cvt_pk_bf16(y.x - __uint_as_float(h.x << 16), y.y - __uint_as_float(h.x & 0xffff0000u));
              l.y = cvt_pk_bf16(y.z - __uint_as_float(h.y << 16), y.w - __uint_as_float(h.y & 0xffff0000u));
              *reinterpret_cast<uint2*>(a.out_hi + (size_t)tok * a.ldop + n) = h;
              *reinterpret_cast<uint2*>(a.out_lo + (size_t)tok * a.ldop + n) = l;
            }
          }
        }
      }
    }
  }
#undef MLP_A
#undef MLP_B
#undef MLP_ACT
#undef MLP_ZERO_HID
#undef MLP_PREFETCH
#undef MLP_INTERLEAVE
#undef MLP_DMA
#undef MLP_STEP_END
}

// Finish of the hidden split: one wave per token row, lane = 4 consecutive columns.  out = (sum of the parts + b2) * gamma + R;
// optional LayerNorm of that row (two-pass statistics over the wave) and / or operand planes, as the fused epilogue.
__global__ __launch_bounds__(256) void k_mlp256_merge(MlpArgs a) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (size_t)a.rows) return;
  const int n = lane * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < a.hsplit; ++s) {
    const float4 p = *reinterpret_cast<const float4*>(a.part + ((size_t)s * a.rows + row) * MD + n);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  if (a.b2) {
    const float4 b = *reinterpret_cast<const float4*>(a.b2 + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (a.gamma) {
    const float4 gm = *reinterpret_cast<const float4*>(a.gamma + n);
    v.x *= gm.x; v.y *= gm.y; v.z *= gm.z; v.w *= gm.w;
  }
  if (a.R) {
    const float4 r = *reinterpret_cast<const float4*>(a.R + row * a.ldr + n);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  *reinterpret_cast<float4*>(a.out + row * a.ldo + n) = v;
  float4 y = v;
  if (a.ln_w) {
    auto wsum = [](float x) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
      return x;
    };
    const float mean = wsum((v.x + v.y) + (v.z + v.w)) * (1.f / MD);
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    const float rstd = 1.f / sqrtf(wsum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.f / MD) + a.ln_eps);
    const float4 w4 = *reinterpret_cast<const float4*>(a.ln_w + n), b4 = *reinterpret_cast<const float4*>(a.ln_b + n);
    y = make_float4(d0 * rstd * w4.x + b4.x, d1 * rstd * w4.y + b4.y, d2 * rstd * w4.z + b4.z, d3 * rstd * w4.w + b4.w);
    if (a.ln_out) *reinterpret_cast<float4*>(a.ln_out + row * a.ldln + n) = y;
  }
  if (a.out_hi) {
    uint2 h, l;
    h.x = cvt_pk_bf16(y.x, y.y);
    h.y = cvt_pk_bf16(y.z, y.w);
    l.x = cvt_pk_bf16(y.x - __uint_as_float(h.x << 16), y.y - __uint_as_float(h.x & 0xffff0000u));
    l.y = cvt_pk_bf16(y.z - __uint_as_float(h.y << 16), y.w - __uint_as_float(h.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(a.out_hi + row * a.ldop + n) = h;
    *reinterpret_cast<uint2*>(a.out_lo + row * a.ldop + n) = l;
  }
}

}  // namespace

// parts of the hidden dimension for `rows` token rows on `ncu` CUs: as many as keep every CU busy, <= 8, each a whole number of
// chunk pairs
int mlp256_hsplit(int rows, int H, int ncu) {
  const int nrb = cdiv(rows, MBR), npair = H / (2 * MHC);
  if (nrb * 2 > ncu) return 1;
  int hs = ncu / nrb;
  if (hs > 8) hs = 8;
  while (hs > 1 && npair % hs) --hs;
  return hs;
}

bool mlp256_supported(const MlpArgs& a) {
  return a.D == MD && a.H % (2 * MHC) == 0 && a.H <= 4096 /* chunk pairs; b1 in LDS */ && a.rows > 0 && a.ldx % 8 == 0 && a.ldw1 % 8 == 0 &&
         a.ldw2 % 8 == 0 && a.ldo % 4 == 0 && (a.R == nullptr || a.ldr % 4 == 0) &&
         (size_t)a.H * a.ldw1 * 2 < (1ull << 32) && (size_t)MD * a.ldw2 * 2 < (1ull << 32);
}

// W2 [n_rows = 256, H] fp32 -> the same matrix with the hidden index permuted inside every group of 16 (fp32; the caller
// splits it into planes).  Once per weight.
int launch_mlp256_permute_w2(const float* w2, int ldw, int n_rows, int H, float* out, hipStream_t st) {
  DS2_REQUIRE(H % 16 == 0, "mlp256: H must be a multiple of 16");
  const size_t n = (size_t)n_rows * H;
  hipLaunchKernelGGL(k_mlp256_permute_w2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w2, ldw, n_rows, H, out);
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

int mlp256_ncu() {
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return ncu;
}
// scratch the hidden split of a launch with `rows` rows needs (0: no split)
size_t mlp256_part_bytes(int rows, int H) {
  const int hs = mlp256_hsplit(rows, H, mlp256_ncu());
  return hs > 1 ? (size_t)hs * rows * MD * sizeof(float) : 0;
}

int launch_mlp256(const MlpArgs& a, hipStream_t st) {
  DS2_REQUIRE(mlp256_supported(a), "mlp256: unsupported shape rows=%d D=%d H=%d", a.rows, a.D, a.H);
  const bool lni = a.lni_w != nullptr;
  DS2_REQUIRE(((a.X_hi && a.X_lo) || lni) && a.W1_hi && a.W1_lo && a.W2_hi && a.W2_lo && a.out, "mlp256: null operand");
  DS2_REQUIRE(!lni || (a.X_f32 && a.lni_b && a.ldxf % 4 == 0 && a.f16x2 && (a.act == DS2_ACT_RELU || a.act == DS2_ACT_GELU)),
              "mlp256: the input LayerNorm is built for the two-fp16-term ReLU / GELU forms (memory attention FFN, memory encoder CXBlock)");
  const int ncu = mlp256_ncu();
  const int nrb = cdiv(a.rows, MBR);
  MlpArgs b = a;
  b.hsplit = 1;
  if (a.part) {
    const int hs = mlp256_hsplit(a.rows, a.H, ncu);
    if (hs > 1 && a.part_bytes >= (size_t)hs * a.rows * MD * sizeof(float)) b.hsplit = hs;
  }
  const int grid = nrb < ncu ? nrb : ncu;
  const size_t smem = (size_t)MNS * MSLOT + (size_t)a.H * 4 + 6 * MD * 4;   // ring + b1 + (LayerNorm weight, bias, b2, gamma; input LayerNorm weight, bias)
  void (*kern)(MlpArgs) = nullptr;
  const bool x2 = a.f16x2 != 0;
  const bool lnf = a.ln_w != nullptr && b.hsplit == 1;   // (with the hidden split the merge kernel normalises)
  DS2_REQUIRE(!lnf || a.act == DS2_ACT_RELU, "mlp256: the LayerNorm epilogue is built for the ReLU form only (memory attention FFN)");
  switch (a.act) {
    case DS2_ACT_NONE: kern = x2 ? k_mlp256<DS2_ACT_NONE, true, false> : k_mlp256<DS2_ACT_NONE, false, false>; break;
    case DS2_ACT_RELU:
      kern = x2 ? (lni ? (lnf ? k_mlp256<DS2_ACT_RELU, true, true, true> : k_mlp256<DS2_ACT_RELU, true, false, true>)
                       : (lnf ? k_mlp256<DS2_ACT_RELU, true, true> : k_mlp256<DS2_ACT_RELU, true, false>))
                : (lnf ? k_mlp256<DS2_ACT_RELU, false, true> : k_mlp256<DS2_ACT_RELU, false, false>);
      break;
    case DS2_ACT_GELU:
      kern = x2 ? (lni ? k_mlp256<DS2_ACT_GELU, true, false, true> : k_mlp256<DS2_ACT_GELU, true, false>) : k_mlp256<DS2_ACT_GELU, false, false>;
      break;
    default: DS2_REQUIRE(false, "mlp256: unsupported activation %d", a.act);
  }
  static bool attr_done[2][2][2][4] = {};
  if (!attr_done[lni][lnf][x2][a.act]) {
    DS2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[lni][lnf][x2][a.act] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid, b.hsplit), dim3(256), smem, st, b);
  DS2_CHECK_LAUNCH();
  if (b.hsplit > 1) {
    hipLaunchKernelGGL(k_mlp256_merge, dim3(cdiv(a.rows, 4)), dim3(256), 0, st, b);
    DS2_CHECK_LAUNCH();
  }
  return DS2_OK;
}
