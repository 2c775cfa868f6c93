// Memory CROSS-attention of mode bf16x3k, assembly form: 4 waves x 64 queries, ONE wave per SIMD, v_mfma_f32_32x32x16_f16, the
// whole key loop one inline-assembly statement with registers allocated by hand (generated: tools/gen/gen_attention_x4a.py ->
// attention_x4a_body.inc; register map and schedule are documented there).
//
// (RoPEAttention.forward of the memory attention's cross_attn_image, sam2/modeling/sam/transformer.py:312-363, in the
// restructured form of DESIGN.md section 4: softmax(Q K^T) M - the values are the raw 64-d memory, v_proj is applied after.)
//
// Round 4 measured every C++ form of this structure slower than the 8-wave kernel (tools/experiments/README_x4.md): with one wave
// per SIMD the step is issue-bound and hipcc neither keeps the 128 registers of Q fragments in the accumulator half nor pipelines
// the K fragment reads.  Pieces:
//   k_x4a_qprep   queries -> rotated (RoPE), scaled by scale * log2(e), fp16, in MFMA B-fragment order (one 16-byte piece per lane)
//   k_vt_pack32   V^T tiles, one fp16 plane, keys permuted to the 32x32 accumulator's row order
//   k_attention_x4a  prologue constants + the assembly loop; writes the UNNORMALISED O^T rows and (max, sum) per query
//   k_w8_merge<64> (attention_w8.hip, nsplit = 1) normalises and writes the bf16 operand planes of the consumer GEMM
#include <stdlib.h>

#include "common.h"
#include <mutex>
#include "kernels.h"
#ifdef X4A_BODY_INC      // (tools/experiments/x4a_bench.hip: schedule experiments)
#include X4A_BODY_INC
#else
#include "attention_x4a_body.inc"
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int DV = 64, BK = 32, KSTEPS = 16;
constexpr int KT_BYTES = BK * 256 * 2, VT_BYTES = DV * BK * 2, RING = 4;

__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {   // v_cvt_pk_f16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{a, b}), f16x2));
}
// key (0..31 inside a tile) -> slot of its value in a V^T row.  The 32x32 accumulator of a lane in half h holds, in register
// r, key (r & 3) + 8 (r >> 2) + 4 h; P.V step s (16 keys of MFMA depth) takes registers 8 s .. 8 s + 7 as the B operand, whose
// k index is 8 h + j.  So slot 16 s + 8 h + j <-> key 16 s + 8 (j >> 2) + 4 h + (j & 3).
__host__ __device__ inline int vt_pos32(int key) { return (key & 16) + 8 * ((key >> 2) & 1) + 4 * ((key >> 3) & 1) + (key & 3); }

// vt[b][tile][dv 0..63][slot 0..31] fp16: one thread per (tile, dv) row (32 key loads coalesced across dv, one 64-byte row store)
__global__ __launch_bounds__(256) void k_vt_pack32(const float* __restrict__ v, int ldv, int batch, int L, unsigned short* __restrict__ vt) {
  const int ntile = (L + 31) / 32;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)batch * ntile * DV) return;
  const int dv = (int)(i % DV);
  const size_t bt = i / DV;
  const int tile = (int)(bt % ntile), b = (int)(bt / ntile);
  const float* src = v + ((size_t)b * L + (size_t)tile * 32) * ldv + dv;
  const int nvalid = L - tile * 32;
  unsigned h[16];
#pragma unroll
  for (int key = 0; key < 32; key += 2) {
    const float x0 = key < nvalid ? src[(size_t)key * ldv] : 0.f;
    const float x1 = key + 1 < nvalid ? src[(size_t)(key + 1) * ldv] : 0.f;
    h[vt_pos32(key) >> 1] = cvt_pk_f16(ds2_sat_f16(x0), ds2_sat_f16(x1));
  }
  uint4* o = reinterpret_cast<uint4*>(vt + (bt * DV + dv) * 32);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) o[q4] = make_uint4(h[4 * q4], h[4 * q4 + 1], h[4 * q4 + 2], h[4 * q4 + 3]);
}

// Q fragments: qfrag[(row block of 64 queries)][qb * 16 + ks][lane] = 8 fp16: lane (q = l31, d = 16 ks + 8 half .. + 7).
// One thread per (64-row block, fragment, lane).
__global__ __launch_bounds__(256) void k_x4a_qprep(const float* __restrict__ q, int ldq, int batch, int Lq, int q_bstride, float sc,
                                                  const float* __restrict__ cis, int rope_grid, int rope_w, uint4* __restrict__ qfrag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nblk64 = (size_t)batch * (Lq / 64);
  if (i >= nblk64 * 32 * 64) return;
  const int lane = (int)(i & 63), frag = (int)((i >> 6) & 31);
  const size_t blk = i >> 11;
  const int b = (int)(blk / (Lq / 64)), r0 = (int)(blk % (Lq / 64)) * 64;
  const int qb = frag >> 4, ks = frag & 15, l31 = lane & 31, half = lane >> 5;
  const int row = r0 + qb * 32 + l31;
  const float* src = q + ((size_t)b * q_bstride + row) * ldq + ks * 16 + half * 8;
  float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
  if (cis) {   // apply_rotary_enc on the complex pairs (d, d + 1): the same expression as k_rope / the 8-wave kernel's query load
    const int t = row % rope_grid, pair0 = ks * 8 + half * 4;
    int tt = t;
    if (rope_w > 0) tt = pair0 < 64 ? t % rope_w : t - t % rope_w;   // pairs < 64 depend on x only, the others on y
    const float4 c0 = *reinterpret_cast<const float4*>(cis + ((size_t)tt * 128 + pair0) * 2);
    const float4 c1 = *reinterpret_cast<const float4*>(cis + ((size_t)tt * 128 + pair0 + 2) * 2);
    v0 = make_float4(v0.x * c0.x - v0.y * c0.y, v0.x * c0.y + v0.y * c0.x, v0.z * c0.z - v0.w * c0.w, v0.z * c0.w + v0.w * c0.z);
    v1 = make_float4(v1.x * c1.x - v1.y * c1.y, v1.x * c1.y + v1.y * c1.x, v1.z * c1.z - v1.w * c1.w, v1.z * c1.w + v1.w * c1.z);
  }
  auto pk = [](float x, float y) { return cvt_pk_f16(ds2_sat_f16(x), ds2_sat_f16(y)); };
  qfrag[i] = make_uint4(pk(v0.x * sc, v0.y * sc), pk(v0.z * sc, v0.w * sc), pk(v1.x * sc, v1.y * sc), pk(v1.z * sc, v1.w * sc));
}

struct X4AArgs {
  const char* k;          // fp16 key plane [batch * Lk][256]
  const char* vt;         // k_vt_pack32 tiles
  const char* qfrag;      // k_x4a_qprep
  float* part_o;          // [nsplit][batch * Lq][64] unnormalised
  float* part_ml;         // [nsplit][batch * Lq][2] (maximum in the log2 domain, sum)
  int batch, Lq, Lk;
  int nsplit;             // key split (few objects): workgroup (x, y) attends key tiles [y * kt_per, ...) - as k_attention_w8
};

__global__ __launch_bounds__(256, 1) void k_attention_x4a(X4AArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[RING * (KT_BYTES + VT_BYTES)];   // 64 KiB K ring + 16 KiB V^T ring
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = a.Lq / 256, nblk = a.batch * nqb;
  int bid = blockIdx.x;
  {   // consecutive block ids (= the query blocks of one object) on one XCD: its K / V^T stay in one L2 (bijective, T1)
    const int xcd = bid % 8, qq = nblk / 8, rr = nblk % 8;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + bid / 8;
  }
  const int b = bid / nqb, q0i = (bid % nqb) * 256;
  const int nkt_all = (a.Lk + BK - 1) / BK;   // keys >= Lk of the last tile: masked to -inf in the loop, V^T rows zero (k_vt_pack32)
  const int kt_per = (nkt_all + a.nsplit - 1) / a.nsplit, kt0 = (int)blockIdx.y * kt_per;
  const int nkt = nkt_all - kt0 < kt_per ? nkt_all - kt0 : kt_per;
  const bool last_part = kt0 + nkt == nkt_all;
  const size_t row0 = (size_t)b * a.Lq + q0i + wave * 64;            // this wave's first query row
  const size_t prow0 = (size_t)blockIdx.y * a.batch * a.Lq + row0;   // ... in this part's slice of the partial results
  const unsigned long long kb = reinterpret_cast<unsigned long long>(a.k + ((size_t)b * a.Lk + (size_t)kt0 * BK) * 512);
  const unsigned long long vb = reinterpret_cast<unsigned long long>(a.vt + ((size_t)b * nkt_all + kt0) * VT_BYTES);
  const unsigned long long qb_ = reinterpret_cast<unsigned long long>(a.qfrag + (row0 / 64) * (32 * 1024));
  const unsigned long long ob = reinterpret_cast<unsigned long long>(a.part_o + prow0 * DV);
  const unsigned long long mb = reinterpret_cast<unsigned long long>(a.part_ml + prow0 * 2);
  const unsigned ldsb = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)lds);
  const unsigned klo = __builtin_amdgcn_readfirstlane((unsigned)kb), khi = __builtin_amdgcn_readfirstlane((unsigned)(kb >> 32));
  const unsigned vlo = __builtin_amdgcn_readfirstlane((unsigned)vb), vhi = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
  const unsigned qlo = __builtin_amdgcn_readfirstlane((unsigned)qb_), qhi = __builtin_amdgcn_readfirstlane((unsigned)(qb_ >> 32));
  const unsigned olo = __builtin_amdgcn_readfirstlane((unsigned)ob), ohi = __builtin_amdgcn_readfirstlane((unsigned)(ob >> 32));
  const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)mb), mhi = __builtin_amdgcn_readfirstlane((unsigned)(mb >> 32));
  const unsigned nval_s = __builtin_amdgcn_readfirstlane((unsigned)(last_part ? a.Lk - (nkt_all - 1) * BK : BK));
  const unsigned nkt_s = __builtin_amdgcn_readfirstlane((unsigned)nkt), ldsb_s = __builtin_amdgcn_readfirstlane(ldsb);
  asm volatile(X4A_ASM_BODY
               :
               : [klo] "s"(klo), [khi] "s"(khi), [vlo] "s"(vlo), [vhi] "s"(vhi), [qlo] "s"(qlo), [qhi] "s"(qhi), [olo] "s"(olo),
                 [ohi] "s"(ohi), [mlo] "s"(mlo), [mhi] "s"(mhi), [nkt] "s"(nkt_s), [nval] "s"(nval_s), [ldsb] "s"(ldsb_s), [wave] "s"(wave), [lane] "v"(lane)
               : X4A_ASM_CLOBBERS);
}

}  // namespace

// (default on; DS2_ATTN_X4A=0 keeps the 8-wave kernel for A/B runs)
bool attention_x4a_enabled() {
  const char* e = getenv("DS2_ATTN_X4A");   // (read per call: the tests compare both kernels in one process)
  const bool on = !(e && atoi(e) == 0);
  return on && DS2_ATTN_K_F16 && ds2_precision() == DS2_PREC_BF16X3K;
}
// key split when the grid of batch * Lq / 256 workgroups leaves CUs idle (few objects): parts of >= 16 key tiles, <= 8 parts
static int x4a_nsplit(int batch, int Lq, int Lk) {
  const int nblk = batch * (Lq / 256), nkt = (Lk + BK - 1) / BK;
  int ns = nblk > 128 ? 1 : 256 / nblk;
  if (ns > 8) ns = 8;
  while (ns > 1 && (nkt + ns - 1) / ns < 16) --ns;
  while (ns > 1 && (ns - 1) * ((nkt + ns - 1) / ns) >= nkt) --ns;   // (no empty last part)
  return ns;
}
// full 256-query blocks.  Lk need not be a multiple of 32, but the key plane must be readable (finite values) up to the end of
// the last tile.
bool attention_x4a_supported(int batch, int Lq, int Lk, int dv, bool planes_out) {
  return dv == DV && planes_out && Lq % 256 == 0 && Lk >= 4 * BK && batch > 0;
}
size_t attention_x4a_ws_bytes(int batch, int Lq, int Lk) {
  const size_t rows = (size_t)batch * Lq;
  return (rows / 64) * (32 * 1024) + (size_t)x4a_nsplit(batch, Lq, Lk) * (rows * DV * sizeof(float) + rows * 2 * sizeof(float)) + 1024;
}

// the same tiles from the bank's bf16 frame entries (kernels.h BankArgs; the fp32 memory tensor is never built): one thread per
// (object, entry, tile of 32 tokens, dv).  A bf16 value converts to fp32 exactly, so the fp16 values equal k_vt_pack32's on `memory`.
__global__ __launch_bounds__(256) void k_bank_vt32(BankArgs a, unsigned short* __restrict__ vt) {
  const int tpe = a.tokens / 32;                       // tiles per entry
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)a.B * a.n_mem * tpe * DV) return;
  const int dv = (int)(i % DV);
  size_t r = i / DV;
  const int tl = (int)(r % tpe); r /= tpe;
  const int e = (int)(r % a.n_mem), b = (int)(r / a.n_mem);
  const unsigned short* src = a.feats[e] + ((size_t)b * a.tokens + (size_t)tl * 32) * 64 + dv;
  unsigned h[16];
#pragma unroll
  for (int key = 0; key < 32; key += 2) {
    const float x0 = __uint_as_float((unsigned)src[(size_t)key * 64] << 16), x1 = __uint_as_float((unsigned)src[(size_t)(key + 1) * 64] << 16);
    h[vt_pos32(key) >> 1] = cvt_pk_f16(ds2_sat_f16(x0), ds2_sat_f16(x1));
  }
  const int ntile = (a.Nk + 31) / 32;
  uint4* o = reinterpret_cast<uint4*>(vt + (((size_t)b * ntile + (size_t)(a.e0 + e) * tpe + tl) * DV + dv) * 32);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) o[q4] = make_uint4(h[4 * q4], h[4 * q4 + 1], h[4 * q4 + 2], h[4 * q4 + 3]);
}
int launch_bank_vt32(const BankArgs& a, void* vt32, hipStream_t st) {
  DS2_REQUIRE(a.n_mem >= 0 && a.n_mem <= DS2_MAX_MEM_ENTRIES && a.tokens % 32 == 0, "bank_vt32: bad entry table (n_mem=%d)", a.n_mem);
  if (a.n_mem > 0) {
    const size_t n = (size_t)a.B * a.n_mem * (a.tokens / 32) * DV;
    hipLaunchKernelGGL(k_bank_vt32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, reinterpret_cast<unsigned short*>(vt32));
    DS2_CHECK_LAUNCH();
  }
  return DS2_OK;
}
const unsigned char* attention_x4a_vt_slot_table() {   // vt_pos32 as a device table (per device, built once; k_bank_ptr_planes)
  static const unsigned char* tab[64] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!tab[dev]) {
    unsigned char h[32];
    for (int k = 0; k < 32; ++k) h[k] = (unsigned char)vt_pos32(k);
    unsigned char* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), 32) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h, 32, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    tab[dev] = d;
  }
  return tab[dev];
}

// the query pass alone (ds2_op_query_fragments: the reference form of gemm_qproj.hip's output)
int launch_x4a_qprep(const float* q, int ldq, int batch, int Lq, bool q_shared, float scale, const float* cis, int rope_grid, void* qfrag,
                     hipStream_t st) {
  DS2_REQUIRE(q && qfrag && ldq % 4 == 0 && batch > 0 && Lq > 0 && Lq % 64 == 0, "x4a_qprep: bad argument");
  int rope_w = 0;
  for (int x = 1; x * x <= rope_grid; ++x)
    if (x * x == rope_grid) rope_w = x;
  const size_t nq = ((size_t)batch * Lq / 64) * 32 * 64;
  hipLaunchKernelGGL(k_x4a_qprep, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, q, ldq, batch, Lq, q_shared ? 0 : Lq,
                     scale * 1.44269504088896340736f, cis, rope_grid, rope_w, reinterpret_cast<uint4*>(qfrag));
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

int launch_vt_pack32(const float* v, int ldv, int batch, int L, void* vt, hipStream_t st) {
  const size_t n = (size_t)batch * ((L + 31) / 32) * DV;
  hipLaunchKernelGGL(k_vt_pack32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, ldv, batch, L, reinterpret_cast<unsigned short*>(vt));
  DS2_CHECK_LAUNCH();
  return DS2_OK;
}

int launch_w8_merge64(const float* part_o, const float* part_ml, int nsplit, size_t rows, void* o_hi, void* o_lo, int ldop, hipStream_t st);

// the un-normalised part(s) of a launch inside its workspace (launch_attention_x4a with merge = false): the consumer combines and
// normalises them itself (launch_vo_merge, gemm_vo.hip).  Returns the number of parts (1 = no key split).
int attention_x4a_parts(void* ws, int batch, int Lq, int Lk, const float** part_o, const float** part_ml) {
  const int ns = x4a_nsplit(batch, Lq, Lk);
  const size_t rows = (size_t)batch * Lq;
  char* w = reinterpret_cast<char*>(ws);
  const float* po = reinterpret_cast<const float*>(w + (rows / 64) * (32 * 1024));
  if (part_o) *part_o = po;
  if (part_ml) *part_ml = po + (size_t)ns * rows * DV;
  return ns;
}

int launch_attention_x4a(const float* q, int ldq, const void* k_f16, const void* vt32, int batch, int Lq, int Lk, float scale,
                         hipStream_t st, void* o_hi, void* o_lo, int ldop, const float* q_rope_cis, int q_rope_grid, bool q_shared,
                         void* ws, size_t ws_bytes, bool merge) {
  DS2_REQUIRE(attention_x4a_supported(batch, Lq, Lk, DV, (o_hi && o_lo) || !merge), "attention_x4a: unsupported shape");
  // q == nullptr: the Q fragments at the start of `ws` are already there (launch_qproj_x4a, gemm_qproj.hip)
  DS2_REQUIRE(ldq % 4 == 0 && ldop % 4 == 0 && k_f16 && vt32 && ws && ws_bytes >= attention_x4a_ws_bytes(batch, Lq, Lk),
              "attention_x4a: bad argument / scratch too small");
  const size_t rows = (size_t)batch * Lq;
  char* w = reinterpret_cast<char*>(ws);
  char* qfrag = w;
  float* part_o = reinterpret_cast<float*>(w + (rows / 64) * (32 * 1024));
  const int nsplit = x4a_nsplit(batch, Lq, Lk);
  float* part_ml = part_o + (size_t)nsplit * rows * DV;
  int rope_w = 0;
  for (int x = 1; x * x <= q_rope_grid; ++x)
    if (x * x == q_rope_grid) rope_w = x;
  DS2_REQUIRE(!q_rope_cis || q_rope_grid > 0, "attention_x4a: rope grid");
  const size_t nq = (rows / 64) * 32 * 64;
  if (q) {
    hipLaunchKernelGGL(k_x4a_qprep, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, q, ldq, batch, Lq, q_shared ? 0 : Lq,
                       scale * 1.44269504088896340736f, q_rope_cis, q_rope_grid, rope_w, reinterpret_cast<uint4*>(qfrag));
    DS2_CHECK_LAUNCH();
  }
  X4AArgs a{reinterpret_cast<const char*>(k_f16), reinterpret_cast<const char*>(vt32), qfrag, part_o, part_ml, batch, Lq, Lk, nsplit};
  hipLaunchKernelGGL(k_attention_x4a, dim3(batch * (Lq / 256), nsplit), dim3(256), 0, st, a);
  DS2_CHECK_LAUNCH();
  if (!merge) return DS2_OK;
  return launch_w8_merge64(part_o, part_ml, nsplit, rows, o_hi, o_lo, ldop, st);
}
