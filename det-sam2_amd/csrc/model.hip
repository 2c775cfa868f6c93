// Stage-level orchestration behind the C-ABI of include/detsam2_hip.h: owns the model parameters,
// the packed/derived weights and one growable device workspace, and strings the gfx950 kernels
// (gemm.hip, attention.hip, kernels.hip) into the reference's stages.  Host code only launches
// kernels on the caller's stream - no host<->device synchronisation on the hot path.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <mutex>
#include <map>
#include <unordered_map>
#include <vector>

#include "../../include/detsam2_hip.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
void ds2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ds2_last_error(void) { return g_err; }
extern "C" int ds2_abi_version(void) { return DS2_ABI_VERSION; }

int g_ds2_default_precision = DS2_PREC_BF16X3K;
thread_local int t_ds2_precision = -1;
extern "C" int ds2_set_precision(int32_t mode) {
  DS2_REQUIRE(mode == DS2_PREC_FP32 || mode == DS2_PREC_BF16X3 || mode == DS2_PREC_BF16X3K,
              "ds2_set_precision: mode must be 0 (fp32), 1 (bf16x3) or 2 (bf16x3k)");
  g_ds2_default_precision = mode;
  return DS2_OK;
}
extern "C" int ds2_get_precision(void) { return g_ds2_default_precision; }

// ------------------------------------------------------------------------------------------------ profiling
// HIP-event brackets on the caller's stream around named launch sites; read back by bench.py for the
// live roofline numbers (ds2_profile_read synchronises the recorded events, never the hot path).
namespace {
struct ProfRec { hipEvent_t a, b; };
bool g_prof = false;
bool g_prof_gemm = false;   // ds2_profile_enable(2): additionally one bracket per GEMM, tagged by shape
std::unordered_map<std::string, std::vector<ProfRec>> g_recs;
struct ProfScope {
  hipStream_t st; std::string tag; ProfRec r; bool on;
  ProfScope(const char* t, hipStream_t s, bool enable = true) : st(s), on(g_prof && enable) {
    if (on) tag = t;
    if (!on) return;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(r.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, st);
    g_recs[tag].push_back(r);
  }
};
}  // namespace
bool ds2_prof_kernels() { return g_prof && g_prof_gemm; }
void ds2_prof_record(const char* tag, hipEvent_t a, hipEvent_t b) { g_recs[tag].push_back(ProfRec{a, b}); }
extern "C" int ds2_profile_enable(int32_t on) { g_prof = on != 0; g_prof_gemm = on == 2; return DS2_OK; }
// newline-separated list of the tags that currently hold records (for per-shape GEMM tables: tags "gemm M N K ...")
extern "C" int ds2_profile_tags(char* buf, int64_t cap) {
  DS2_REQUIRE(buf && cap > 0, "ds2_profile_tags: bad argument");
  std::string all;
  for (auto& kv : g_recs)
    if (!kv.second.empty()) { all += kv.first; all += '\n'; }
  DS2_REQUIRE((int64_t)all.size() < cap, "ds2_profile_tags: buffer too small (%zu bytes needed)", all.size() + 1);
  memcpy(buf, all.c_str(), all.size() + 1);
  return DS2_OK;
}
extern "C" int ds2_profile_read(const char* tag, double* total_ms, int64_t* launches) {
  DS2_REQUIRE(tag && total_ms && launches, "ds2_profile_read: null argument");
  *total_ms = 0.0; *launches = 0;
  auto it = g_recs.find(tag);
  if (it == g_recs.end()) return DS2_OK;
  for (ProfRec& r : it->second) {
    float ms = 0.f;
    DS2_CHECK_HIP(hipEventSynchronize(r.b));
    DS2_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    *total_ms += ms; *launches += 1;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  it->second.clear();
  return DS2_OK;
}

#define TRY(x)              \
  do {                      \
    int _r = (x);           \
    if (_r != DS2_OK) return _r; \
  } while (0)

// ------------------------------------------------------------------------------------------------ model
namespace {

struct Blob { void* ptr = nullptr; size_t bytes = 0; };

struct BlockCfg { int dim, dim_out, heads, window, q_stride; };

constexpr int TOK = 4096;  // 64x64 tokens at stride 16

}  // namespace

// bf16x3 operand planes of weights (split once, cached) and the scratch for activation planes that no producer emitted:
// owned by the model (two predictors in one process - other GPUs, other streams - must not share them); the primitive
// ops (ds2_op_gemm), which have no model, use one process-wide context.
struct GemmPlanes { unsigned short *hi = nullptr, *lo = nullptr; int ld = 0; };
struct GemmCtx {
  typedef std::unordered_map<const float*, GemmPlanes> PlaneMap;
  PlaneMap own_wcache, own_w2perm, own_wcache16, own_w2perm16;   // (..16: IEEE fp16 planes of the two-term products)
  PlaneMap own_wcache_mx;                                         // "MX" weight planes of the two-MFMA-equivalent product (common.h)
  PlaneMap& wcmx() { return share ? share->own_wcache_mx : own_wcache_mx; }
  std::unordered_map<const float*, float*> own_wt;   // skinny linear layers: fp32 weights transposed to [K, N] (once)
  std::unordered_map<const float*, float*>& wt() { return share ? share->own_wt : own_wt; }
  GemmCtx* share = nullptr;   // a VIEW model (ds2_model_create_view) uses its parent's weight planes; the scratch is its own
  // the three maps are shared by a parent and its views, which may be driven from different host threads: every lookup /
  // first-use insertion runs under the OWNER's mutex (held across the split + publish, so a second thread finds finished planes)
  std::mutex own_mu;
  std::mutex& mu() { return share ? share->own_mu : own_mu; }
  PlaneMap& wc() { return share ? share->own_wcache : own_wcache; }        // weights split into planes (once)
  PlaneMap& w2p() { return share ? share->own_w2perm : own_w2perm; }      // fused MLP: W2 with the hidden index permuted
  PlaneMap& wc16() { return share ? share->own_wcache16 : own_wcache16; }
  PlaneMap& w2p16() { return share ? share->own_w2perm16 : own_w2perm16; }
  // a weight's planes may be consumed on another stream than the one that split it (view models): the creating call waits
  // for its split kernels once (first use only)
  int publish(hipStream_t st) { DS2_CHECK_HIP(hipStreamSynchronize(st)); return DS2_OK; }
  char* scratch = nullptr;
  size_t scratch_cap = 0;
  int require(size_t bytes, hipStream_t st) {
    if (bytes <= scratch_cap) return DS2_OK;
    DS2_CHECK_HIP(hipStreamSynchronize(st));
    if (scratch) DS2_CHECK_HIP(hipFree(scratch));
    scratch = nullptr; scratch_cap = 0;
    const size_t want = bytes + bytes / 4 + (16u << 20);
    DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&scratch), want));
    scratch_cap = want;
    return DS2_OK;
  }
  void release() {
    for (auto& kv : own_wcache) { (void)hipFree(kv.second.hi); (void)hipFree(kv.second.lo); }
    own_wcache.clear();
    for (auto& kv : own_w2perm) { (void)hipFree(kv.second.hi); (void)hipFree(kv.second.lo); }
    own_w2perm.clear();
    for (PlaneMap* mp : {&own_wcache16, &own_w2perm16, &own_wcache_mx}) {
      for (auto& kv : *mp) { (void)hipFree(kv.second.hi); (void)hipFree(kv.second.lo); }
      mp->clear();
    }
    for (auto& kv : own_wt) (void)hipFree(kv.second);
    own_wt.clear();
    if (scratch) (void)hipFree(scratch);
    scratch = nullptr; scratch_cap = 0;
  }
};
static GemmCtx g_gemm_ctx;   // primitives only

// Every model entry point runs with the model's device current (allocations, launches) and restores the caller's.
struct DeviceGuard {
  int prev = -1, dev;
  explicit DeviceGuard(int d) : dev(d) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};

struct ds2_model;
static int model_precision(const ds2_model* m);
// Every model entry point: the model's device current + the model's arithmetic mode installed for this thread.
struct ModelScope {
  DeviceGuard dg;
  ds2_precision_scope ps;
  explicit ModelScope(const ds2_model* m);
};

struct ds2_model {
  ds2_config cfg;
  int device = 0;              // the device that was current at ds2_model_create
  int precision = DS2_PREC_BF16X3K;   // arithmetic mode of this model's stages (ds2_model_set_precision)
  bool ma_fold_vo = false;            // memory attention: out_proj folded into the value projections (set at finalize)
  ds2_model* parent = nullptr;        // VIEW of another model (ds2_model_create_view): its parameters and weight planes, own workspace
  GemmCtx gctx;
  std::vector<BlockCfg> blocks;
  std::vector<int> stage_ends;
  std::unordered_map<std::string, Blob> params;
  bool finalized = false;
  // workspace arena (bump allocator, reset per stage)
  char* ws = nullptr;
  size_t ws_cap = 0, ws_top = 0;
  std::string missing;  // first missing parameter seen by P()
  // bf16x3 operand planes of activations living in the arena, keyed by the fp32 buffer they stand for
  // (producers that emit planes directly register them here; gemm() looks its A operand up)
  struct ActPlanes { unsigned short *hi, *lo; int ld; int fmt = DS2_PLANES_BF16; };   // fmt: common.h (bf16 hi / lo | "MX")
  std::unordered_map<const void*, ActPlanes> act_planes;
  void release(size_t mark) {   // rewind the arena and forget planes of buffers above the mark
    for (auto it = act_planes.begin(); it != act_planes.end();)
      it = (reinterpret_cast<const char*>(it->first) >= ws + mark) ? act_planes.erase(it) : std::next(it);
    ws_top = mark;
  }

  const float* P(const std::string& name) {
    if (parent) {
      const float* p = parent->P(name);
      if (!p && missing.empty()) missing = name;
      return p;
    }
    auto it = params.find(name);
    if (it == params.end()) {
      if (missing.empty()) missing = name;
      return nullptr;
    }
    return reinterpret_cast<const float*>(it->second.ptr);
  }
  size_t Pbytes(const std::string& name) {
    if (parent) return parent->Pbytes(name);
    auto it = params.find(name);
    return it == params.end() ? 0 : it->second.bytes;
  }
  float* alloc(size_t n_floats) { return reinterpret_cast<float*>(alloc_bytes(n_floats * sizeof(float))); }
  void* alloc_bytes(size_t bytes) {
    const size_t aligned = (bytes + 255) & ~(size_t)255;
    if (ws_top + aligned > ws_cap) return nullptr;
    void* p = ws + ws_top;
    ws_top += aligned;
    return p;
  }
  int require(size_t bytes, hipStream_t st) {   // (re)size the arena; only syncs when it has to grow
    ws_top = 0;
    act_planes.clear();
    if (bytes <= ws_cap) return DS2_OK;
    DS2_CHECK_HIP(hipStreamSynchronize(st));
    if (ws) DS2_CHECK_HIP(hipFree(ws));
    ws = nullptr;
    ws_cap = 0;
    const size_t want = bytes + bytes / 8 + (64u << 20);
    DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ws), want));
    ws_cap = want;
    return DS2_OK;
  }
  int add_derived(const std::string& name, size_t n_floats, float** out) {
    if (parent) return parent->add_derived(name, n_floats, out);
    Blob b;
    b.bytes = n_floats * sizeof(float);
    DS2_CHECK_HIP(hipMalloc(&b.ptr, b.bytes));
    params[name] = b;
    *out = reinterpret_cast<float*>(b.ptr);
    return DS2_OK;
  }
};

static int model_precision(const ds2_model* m) { return m->precision; }
ModelScope::ModelScope(const ds2_model* m) : dg(m->device), ps(m->precision) {}

#define ALLOC(var, n)                                                        \
  float* var = m->alloc(n);                                                  \
  if (!var) {                                                                \
    ds2_set_error("%s:%d workspace exhausted allocating %zu floats (cap %zu)", __FILE__, __LINE__, (size_t)(n), m->ws_cap); \
    return DS2_ERR_STATE;                                                    \
  }

#define CHECK_PARAMS()                                                          \
  if (!m->missing.empty()) {                                                    \
    ds2_set_error("missing parameter '%s'", m->missing.c_str());                \
    m->missing.clear();                                                         \
    return DS2_ERR_STATE;                                                       \
  }

// ---- bf16x3 operand planes: weights are split once and cached (model-owned pointers only); activations are
// split per call into a process-wide scratch buffer (until their producers emit planes directly).
namespace {
inline int round32(int k) { return (k + 31) / 32 * 32; }
}  // namespace

static int round32i(int k) { return (k + 31) / 32 * 32; }
// allocate + register the planes that stand for fp32 buffer `key` ([rows, cols]); pad columns are zeroed
static int new_act_planes(ds2_model* m, const void* key, int rows, int cols, ds2_model::ActPlanes* out, hipStream_t st) {
  const int ld = round32i(cols);
  const size_t bytes = (size_t)rows * ld * 2;
  out->hi = reinterpret_cast<unsigned short*>(m->alloc_bytes(bytes));
  out->lo = reinterpret_cast<unsigned short*>(m->alloc_bytes(bytes));
  out->ld = ld;
  if (!out->hi || !out->lo) { ds2_set_error("workspace exhausted allocating activation planes"); return DS2_ERR_STATE; }
  if (ld != cols) {   // producers only write the real columns
    DS2_CHECK_HIP(hipMemsetAsync(out->hi, 0, bytes, st));
    DS2_CHECK_HIP(hipMemsetAsync(out->lo, 0, bytes, st));
  }
  m->act_planes[key] = *out;
  return DS2_OK;
}

// C = act(A W^T + bias) * gamma + R.   bf16x3 mode: A is taken from registered planes when its producer emitted
// them (else split by a pre-pass); with planes_out the result is emitted as planes registered under key C and
// the fp32 buffer C is NOT written (its only consumers must be GEMMs).
// Would gemm() run this Linear layer as the two-MFMA-equivalent ("MX") product?  Shape rules of the assembly kernel's 128 x 192
// configuration (gemm_x4g.hip) - NOT its chip-filling rule: the arithmetic of a layer must not depend on the batch size (a sharded
// stream encodes other batch sizes than a sequential one).  DS2_GEMM_MX=0: the three-term bf16 product everywhere (A/B runs).
static bool gemm_mx_wanted(int M, int N, int K, int act, bool has_r, int r_mod, bool has_gamma, bool has_rope, bool planes_out,
                           bool out_hi_only, bool has_bias, bool w_static) {
  (void)w_static;
  if (ds2_precision() != DS2_PREC_BF16X3K || !has_bias || has_gamma || has_rope || r_mod != 0 || out_hi_only) return false;
  const char* e = getenv("DS2_GEMM_MX");      // (read per call: the tests compare both products in one process)
  if (e && atoi(e) == 0) return false;
  if (M % 128 || N % 192 || K % 64 || K < 576) return false;
  if ((unsigned long long)M * K * 2 >= (1ull << 32) || (unsigned long long)M * N * 4 >= (1ull << 32)) return false;   // 32-bit offsets inside an operand
  if (planes_out) return act == DS2_ACT_GELU && !has_r;                  // epilogue forms e2 | e1 | e3 of the assembly kernel
  return act == DS2_ACT_NONE;
}
static int new_act_planes(ds2_model* m, const void* key, int rows, int cols, ds2_model::ActPlanes* out, hipStream_t st);
static int gemm_mx(hipStream_t st, int M, int N, int K, int Kp, const float* A, int lda, const unsigned short* ahi, const unsigned short* alo,
                   int a_fmt, const float* W, int ldw, const float* bias, float* C, int ldc, int act, const float* R, int ldr, ds2_model* m,
                   bool planes_out, bool w_static, bool out_mx);
static int gemm(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias,
                float* C, int ldc, int act = DS2_ACT_NONE, const float* R = nullptr, int ldr = 0, int r_mod = 0,
                const float* gamma = nullptr, bool w_static = false, ds2_model* m = nullptr, bool planes_out = false,
                const float* rope_cis = nullptr, int rope_L = 0, int rope_n = 0, int rope_grid = 0, bool out_hi_only = false,
                bool out_hi_f16 = false, bool out_mx = false) {
  if (!A || !W || !C) {
    ds2_set_error("gemm: null operand (missing parameter?)");
    return DS2_ERR_STATE;
  }
  // a handful of rows against a model weight (token side of the two-way transformer, small heads): spread over the chip in
  // exact fp32 instead of one or two latency-bound tiles (gemm_skinny.hip)
  if (M <= 128 && w_static && m && !planes_out && !rope_cis && K % 4 == 0 && lda % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(A) & 15) == 0 && !m->act_planes.count(A)) {
    char ptag[96] = "";
    if (g_prof_gemm) snprintf(ptag, sizeof(ptag), "gemm %d %d %d", M, N, K);
    ProfScope _gp(ptag, st, g_prof_gemm);
    GemmCtx& ctx = m->gctx;
    float* wt = nullptr;
    std::unique_lock<std::mutex> lk(ctx.mu());
    auto it = ctx.wt().find(W);
    if (it != ctx.wt().end()) {
      wt = it->second;
    } else {
      DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&wt), (size_t)N * K * 4));
      TRY(launch_transpose_w(W, ldw, N, K, wt, st));
      TRY(ctx.publish(st));
      ctx.wt()[W] = wt;
    }
    lk.unlock();
    SkinnyArgs g{M, N, K, A, lda, wt, bias, gamma, R, ldr, r_mod, C, ldc, act};
    return launch_skinny_linear(g, st);
  }
  if (!ds2_split_mode()) {
    GemmArgs g{M, N, K, A, lda, W, ldw, bias, C, ldc, act, gamma, R, ldr, r_mod};
    return launch_gemm(g, st);
  }
  DS2_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "gemm: K/lda/ldw must be multiples of 4");
  // per-shape timing (ds2_profile_enable(2)): HIP events around the whole GEMM incl. any operand split pre-pass
  char ptag[96] = "";
  if (g_prof_gemm) snprintf(ptag, sizeof(ptag), "gemm %d %d %d", M, N, K);
  ProfScope _gp(ptag, st, g_prof_gemm);
  const int Kp = round32(K);
  const size_t w_bytes = (size_t)N * Kp * 2;
  const unsigned short *ahi = nullptr, *alo = nullptr;
  int a_ld = Kp, a_fmt = DS2_PLANES_BF16;
  if (m) {
    auto ia = m->act_planes.find(A);
    if (ia != m->act_planes.end() && ia->second.ld == Kp) { ahi = ia->second.hi; alo = ia->second.lo; a_fmt = ia->second.fmt; }
  }
  // the two-MFMA-equivalent product (mode bf16x3k; gemm_x4g.hip "23m", common.h "MX" planes): the Linear layers whose shape the
  // assembly kernel's 128 x 192 configuration takes - Hiera stages 3 / 4 of hiera_l (hieradet.py:40-82,86-168), whatever the batch
  const bool mx = (!planes_out || out_mx) &&      // (the MX form's plane epilogue writes MX planes: only for an MX consumer)
                  gemm_mx_wanted(M, N, K, act, R != nullptr, r_mod, gamma != nullptr, rope_cis != nullptr, planes_out, out_hi_only, bias != nullptr, w_static) &&
                  (!C || ((reinterpret_cast<uintptr_t>(C) & 15) == 0 && ldc % 4 == 0)) && (!R || ((reinterpret_cast<uintptr_t>(R) & 15) == 0 && ldr % 4 == 0)) &&
                  (reinterpret_cast<uintptr_t>(bias) & 15) == 0;
  if (mx) return gemm_mx(st, M, N, K, Kp, A, lda, ahi, alo, a_fmt, W, ldw, bias, C, ldc, act, R, ldr, m, planes_out, w_static, out_mx);
  DS2_REQUIRE(a_fmt == DS2_PLANES_BF16 && !out_mx, "gemm: MX operand planes reached a GEMM that does not multiply in the MX form (M=%d N=%d K=%d)", M, N, K);
  const size_t a_bytes = ahi ? 0 : (size_t)M * Kp * 2;
  GemmCtx& ctx = m ? m->gctx : g_gemm_ctx;
  GemmPlanes wp;
  std::unique_lock<std::mutex> lk(ctx.mu());
  auto it = w_static ? ctx.wc().find(W) : ctx.wc().end();
  if (it != ctx.wc().end()) {
    wp = it->second;
    TRY(ctx.require(2 * a_bytes + 512, st));
  } else if (w_static) {
    DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&wp.hi), w_bytes));
    DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&wp.lo), w_bytes));
    wp.ld = Kp;
    TRY(launch_split_rows(W, ldw, N, K, wp.hi, wp.lo, Kp, st));
    TRY(ctx.publish(st));
    ctx.wc()[W] = wp;
    TRY(ctx.require(2 * a_bytes + 512, st));
  } else {
    TRY(ctx.require(2 * a_bytes + 2 * w_bytes + 1024, st));
    wp.hi = reinterpret_cast<unsigned short*>(ctx.scratch + ((2 * a_bytes + 255) & ~(size_t)255));
    wp.lo = wp.hi + (size_t)N * Kp;
    wp.ld = Kp;
    TRY(launch_split_rows(W, ldw, N, K, wp.hi, wp.lo, Kp, st));
  }
  lk.unlock();
  if (!ahi) {
    unsigned short* sh = reinterpret_cast<unsigned short*>(ctx.scratch);
    unsigned short* sl = sh + (size_t)M * Kp;
    TRY(launch_split_rows(A, lda, M, K, sh, sl, Kp, st));
    ahi = sh; alo = sl;
  }
  GemmSplitArgs g{};
  g.M = M; g.N = N; g.Kp = Kp;
  g.A_hi = ahi; g.A_lo = alo; g.lda = a_ld;
  g.W_hi = wp.hi; g.W_lo = wp.lo; g.ldw = wp.ld;
  g.bias = bias; g.C = C; g.ldc = ldc; g.act = act; g.gamma = gamma; g.R = R; g.ldr = ldr; g.r_mod = r_mod;
  if (planes_out && m) {
    ds2_model::ActPlanes op;
    TRY(new_act_planes(m, C, M, N, &op, st));
    g.C = nullptr; g.C_hi = op.hi; g.C_lo = out_hi_only ? nullptr : op.lo; g.ldcp = op.ld;
    g.c_hi_f16 = (out_hi_only && out_hi_f16) ? 1 : 0;
    g.rope_cis = rope_cis; g.rope_L = rope_L; g.rope_n = rope_n; g.rope_grid = rope_grid;
    if (rope_cis && m->cfg.image_size / 16 * (m->cfg.image_size / 16) == rope_grid) g.rope_w = m->cfg.image_size / 16;
  }
  return launch_gemm_split(g, st);
}
// gemm() in the MX form.  The weight's MX planes are built once and cached; the activation's come from its registered bf16 planes
// (a = hi + lo to 2^-17) or from the fp32 tensor by a pre-pass.
static int gemm_mx(hipStream_t st, int M, int N, int K, int Kp, const float* A, int lda, const unsigned short* ahi, const unsigned short* alo,
                   int a_fmt, const float* W, int ldw, const float* bias, float* C, int ldc, int act, const float* R, int ldr, ds2_model* m,
                   bool planes_out, bool w_static, bool out_mx) {
  DS2_REQUIRE(Kp == K, "gemm_mx: K must be a multiple of 64");
  GemmCtx& ctx = m ? m->gctx : g_gemm_ctx;
  GemmPlanes wp;
  const size_t a_bytes = (size_t)M * Kp * 2, w_bytes = (size_t)N * Kp * 2;
  {
    std::lock_guard<std::mutex> lk(ctx.mu());
    auto it = w_static ? ctx.wcmx().find(W) : ctx.wcmx().end();
    if (it != ctx.wcmx().end()) {
      wp = it->second;
    } else if (!w_static) {      // (model-less primitive: the weight's planes behind the activation's in the scratch)
      TRY(ctx.require(2 * a_bytes + 2 * w_bytes + 1024, st));
      wp.hi = reinterpret_cast<unsigned short*>(ctx.scratch + ((2 * a_bytes + 255) & ~(size_t)255));
      wp.lo = wp.hi + (size_t)N * Kp;
      wp.ld = Kp;
      TRY(launch_split_rows(W, ldw, N, K, wp.hi, wp.lo, Kp, st, false, DS2_PLANES_MX_W));
    } else {
      DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&wp.hi), (size_t)N * Kp * 2));
      DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&wp.lo), (size_t)N * Kp * 2));
      wp.ld = Kp;
      TRY(launch_split_rows(W, ldw, N, K, wp.hi, wp.lo, Kp, st, false, DS2_PLANES_MX_W));
      TRY(ctx.publish(st));
      ctx.wcmx()[W] = wp;
    }
    TRY(ctx.require(2 * a_bytes + 512, st));
  }
  const unsigned short *p1 = ahi, *p2 = alo;
  if (!ahi || a_fmt != DS2_PLANES_MX_A) {     // no MX planes from the producer: a pre-pass builds them in the scratch
    unsigned short* s1 = reinterpret_cast<unsigned short*>(ctx.scratch);
    unsigned short* s2 = s1 + (size_t)M * Kp;
    if (ahi) TRY(launch_planes_bf16_to_mx(ahi, alo, s1, s2, (size_t)M * Kp, st));
    else TRY(launch_split_rows(A, lda, M, K, s1, s2, Kp, st, false, DS2_PLANES_MX_A));
    p1 = s1; p2 = s2;
  }
  GemmSplitArgs g{};
  g.M = M; g.N = N; g.Kp = Kp;
  g.A_hi = p1; g.A_lo = p2; g.lda = Kp;
  g.W_hi = wp.hi; g.W_lo = wp.lo; g.ldw = wp.ld;
  g.bias = bias; g.C = C; g.ldc = ldc; g.act = act; g.R = R; g.ldr = ldr;
  g.mx = 1;
  if (planes_out && m) {
    ds2_model::ActPlanes op;
    TRY(new_act_planes(m, C, M, N, &op, st));
    g.C = nullptr; g.C_hi = op.hi; g.C_lo = op.lo; g.ldcp = op.ld;
    if (out_mx) { g.c_mx = 1; m->act_planes[C].fmt = DS2_PLANES_MX_A; }
  }
  return launch_gemm_split(g, st);
}
// Linear layer by state_dict prefix: y = act(x W^T + b) (+ R)
static int linear(ds2_model* m, hipStream_t st, const std::string& p, int M, int N, int K, const float* A, int lda,
                  float* C, int ldc, int act = DS2_ACT_NONE, const float* R = nullptr, int ldr = 0, int r_mod = 0,
                  const float* gamma = nullptr, bool planes_out = false, bool out_mx = false) {
  return gemm(st, M, N, K, A, lda, m->P(p + ".weight"), K, m->P(p + ".bias"), C, ldc, act, R, ldr, r_mod, gamma, true, m,
              planes_out, nullptr, 0, 0, 0, false, false, out_mx);
}
// Will linear() multiply this layer in the MX form?  (producers of its A operand ask, to emit the planes in that format)
static bool linear_mx(ds2_model* m, const std::string& p, int M, int N, int K, int act, bool has_r, bool planes_out) {
  const float* b = m->P(p + ".bias");
  return gemm_mx_wanted(M, N, K, act, has_r, 0, false, false, planes_out, false, b != nullptr, true) && (reinterpret_cast<uintptr_t>(b) & 15) == 0;
}
// weight planes of a static weight [N, K] (split once, cached)
static int weight_planes(GemmCtx& ctx, const float* W, int N, int K, GemmPlanes* out, hipStream_t st, bool f16 = false) {
  std::lock_guard<std::mutex> lk(ctx.mu());
  GemmCtx::PlaneMap& map = f16 ? ctx.wc16() : ctx.wc();
  auto it = map.find(W);
  if (it != map.end()) { *out = it->second; return DS2_OK; }
  const int Kp = round32i(K);
  GemmPlanes wp;
  DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&wp.hi), (size_t)N * Kp * 2));
  DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&wp.lo), (size_t)N * Kp * 2));
  wp.ld = Kp;
  TRY(launch_split_rows(W, K, N, K, wp.hi, wp.lo, Kp, st, f16));
  TRY(ctx.publish(st));
  map[W] = wp;
  *out = wp;
  return DS2_OK;
}
// two-term fp16 products in the memory attention / memory encoder of mode bf16x3k (gemm_mlp256.hip X2; DS2_F16X2=0: three bf16 terms)
static bool f16x2_enabled() {
  const char* e = getenv("DS2_F16X2");   // (read per call: the tests compare both in one process)
  return !(e && atoi(e) == 0) && ds2_precision() == DS2_PREC_BF16X3K;
}
// LayerNorm fused into the fused MLP's epilogue (gemm_mlp256.hip): weights, and where LN(result) goes
struct LnFuse {
  const float *w, *b; float eps;
  float* out_f32;                    // fp32 [rows, 256] (nullable)
  const ds2_model::ActPlanes* planes;   // pre-allocated operand planes, ld 256 (nullable)
};
// Scratch of the model-less fused-MLP op (ds2_op_mlp): one buffer per (host thread, device, purpose).  A call on another stream than the
// previous one first waits for that stream - the buffer may still be read by its kernels; growing frees the old buffer
// (hipFree waits for the device).
static char* op_scratch(int kind, size_t need, hipStream_t st) {
  struct Buf { char* p = nullptr; size_t bytes = 0; hipStream_t last = nullptr; bool used = false; };
  static thread_local std::map<std::pair<int, int>, Buf> bufs;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  Buf& b = bufs[{dev, kind}];
  if (b.used && b.last != st && hipStreamSynchronize(b.last) != hipSuccess) return nullptr;
  if (b.bytes < need) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.bytes = 0;
    if (hipMalloc(reinterpret_cast<void**>(&b.p), need) != hipSuccess) return nullptr;
    b.bytes = need;
  }
  b.last = st; b.used = true;
  return b.p;
}
// out = (act(A W1^T + b1) W2^T + b2) * gamma + R with the hidden activations kept in registers (gemm_mlp256.hip); bf16x3
// modes, model width 256 only - the caller falls back to two GEMMs when this returns DS2_ERR_UNSUPPORTED.
// A's operand planes must be registered (its producer emitted them) or are split here.
static int mlp_fused(ds2_model* m, GemmCtx& ctx, hipStream_t st, int rows, int H, const float* A, int lda, const float* W1,
                     const float* b1, const float* W2, const float* b2, const float* gamma, const float* R, int ldr, float* out,
                     int ldo, int act, bool planes_too = false, bool f16x2 = false, const LnFuse* lnf = nullptr,
                     const LnFuse* lni = nullptr) {   // lni: LayerNorm of the INPUT rows in the kernel's prologue (A = the un-normalised fp32 rows)
  if (!ds2_split_mode() || !A || !W1 || !W2 || !out) return DS2_ERR_UNSUPPORTED;
  MlpArgs a{};
  a.f16x2 = f16x2 ? 1 : 0;
  if (lnf) {   // LayerNorm of the result rows in the epilogue: planes (and / or fp32) of LN(result)
    a.ln_w = lnf->w; a.ln_b = lnf->b; a.ln_eps = lnf->eps; a.ln_out = lnf->out_f32; a.ldln = 256;
    if (lnf->planes) { a.out_hi = lnf->planes->hi; a.out_lo = lnf->planes->lo; a.ldop = lnf->planes->ld; }
  }
  a.rows = rows; a.D = 256; a.H = H; a.ldx = 256; a.ldw1 = 256; a.ldw2 = H;
  a.b1 = b1; a.b2 = b2; a.gamma = gamma; a.R = R; a.ldr = ldr; a.out = out; a.ldo = ldo; a.act = act;
  if (!mlp256_supported(a) || !(act == DS2_ACT_NONE || act == DS2_ACT_RELU || act == DS2_ACT_GELU)) return DS2_ERR_UNSUPPORTED;
  char ptag[96] = "";
  if (g_prof_gemm) snprintf(ptag, sizeof(ptag), "kern k_mlp256 %d %d %d", rows, 256, 2 * H);   // (2*M*N*K with K = 2H: both layers)
  GemmPlanes w1p, w2p;
  if (!m) {
    // primitive call (ds2_op_mlp): the caller owns its weight tensors and may free / reuse their addresses between calls, so nothing
    // is cached under them - the planes of both weights are rebuilt into a per-thread scratch on the caller's stream (the
    // earlier insert-then-forget paid a device synchronisation, hipFree and four hipMallocs per call)
    const size_t pl = (size_t)H * 256 * 2, need = 4 * pl + (size_t)256 * H * 4;
    char* op_w = op_scratch(0, need, st);
    if (!op_w) { ds2_set_error("ds2_op_mlp: scratch allocation failed"); return DS2_ERR_HIP; }
    w1p.hi = reinterpret_cast<unsigned short*>(op_w); w1p.lo = reinterpret_cast<unsigned short*>(op_w + pl); w1p.ld = 256;
    w2p.hi = reinterpret_cast<unsigned short*>(op_w + 2 * pl); w2p.lo = reinterpret_cast<unsigned short*>(op_w + 3 * pl); w2p.ld = H;
    float* tmp = reinterpret_cast<float*>(op_w + 4 * pl);
    TRY(launch_split_rows(W1, 256, H, 256, w1p.hi, w1p.lo, 256, st, f16x2));
    TRY(launch_mlp256_permute_w2(W2, H, 256, H, tmp, st));
    TRY(launch_split_rows(tmp, H, 256, H, w2p.hi, w2p.lo, H, st, f16x2));
  } else {
    TRY(weight_planes(ctx, W1, H, 256, &w1p, st, f16x2));
    std::unique_lock<std::mutex> lk2(ctx.mu());
    GemmCtx::PlaneMap& map2 = f16x2 ? ctx.w2p16() : ctx.w2p();
    auto it = map2.find(W2);
    if (it != map2.end()) {
      w2p = it->second;
    } else {   // once per weight: permute the hidden index inside groups of 16, then split
      float* tmp = nullptr;
      DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)256 * H * 4));
      TRY(launch_mlp256_permute_w2(W2, H, 256, H, tmp, st));
      DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&w2p.hi), (size_t)256 * H * 2));
      DS2_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&w2p.lo), (size_t)256 * H * 2));
      w2p.ld = H;
      TRY(launch_split_rows(tmp, H, 256, H, w2p.hi, w2p.lo, H, st, f16x2));
      DS2_CHECK_HIP(hipStreamSynchronize(st));
      DS2_CHECK_HIP(hipFree(tmp));
      map2[W2] = w2p;
    }
    lk2.unlock();
  }
  const unsigned short *xh = nullptr, *xl = nullptr;
  if (lni) {
    DS2_REQUIRE(f16x2 && (act == DS2_ACT_RELU || act == DS2_ACT_GELU) && lni->w && lni->b && lda % 4 == 0,
                "mlp_fused: input LayerNorm needs the two-fp16-term ReLU / GELU form");
    a.X_f32 = A; a.ldxf = lda; a.lni_w = lni->w; a.lni_b = lni->b; a.lni_eps = lni->eps;
  } else if (m) {
    auto ia = m->act_planes.find(A);
    if (ia != m->act_planes.end() && ia->second.ld == 256) { xh = ia->second.hi; xl = ia->second.lo; }
  }
  if (!xh && !lni) {
    TRY(ctx.require((size_t)rows * 256 * 4 + 512, st));
    unsigned short* sh = reinterpret_cast<unsigned short*>(ctx.scratch);
    unsigned short* sl = sh + (size_t)rows * 256;
    TRY(launch_split_rows(A, lda, rows, 256, sh, sl, 256, st));
    xh = sh; xl = sl;
  }
  a.X_hi = xh; a.X_lo = xl;
  if (const size_t pb = mlp256_part_bytes(rows, H)) {   // few rows: hidden dimension split over workgroups, parts merged after
    a.part = reinterpret_cast<float*>(m ? m->alloc_bytes(pb) : nullptr);
    if (!a.part && !m) {   // primitive call: a scratch of its own behind the operand split
      a.part = reinterpret_cast<float*>(op_scratch(1, pb, st));
      if (!a.part) { ds2_set_error("ds2_op_mlp: scratch allocation failed"); return DS2_ERR_HIP; }
    }
    a.part_bytes = a.part ? pb : 0;
  }
  if (planes_too && !lnf && m && ldo == 256) {   // the result also as the next GEMM's operand planes, registered under `out`
    ds2_model::ActPlanes op;
    TRY(new_act_planes(m, out, rows, 256, &op, st));
    a.out_hi = op.hi; a.out_lo = op.lo; a.ldop = op.ld;
  }
  a.W1_hi = w1p.hi; a.W1_lo = w1p.lo; a.ldw1 = w1p.ld;
  a.W2_hi = w2p.hi; a.W2_lo = w2p.lo; a.ldw2 = w2p.ld;
  ProfScope _gp(ptag, st, g_prof_gemm);
  return launch_mlp256(a, st);
}
// two-layer MLP by state_dict prefixes: fused when the shape allows it, else the two GEMMs (hidden planes in `hbuf`)
// planes_too: `out` feeds a GEMM next - the fused kernel writes its operand planes alongside the fp32 result
static int mlp2(ds2_model* m, hipStream_t st, const std::string& p1, const std::string& p2, int rows, int H, const float* A,
                float* hbuf, float* out, int act, const float* R, const float* gamma, bool planes_too = false, bool f16x2 = false,
                const LnFuse* lnf = nullptr, bool* ln_done = nullptr, const LnFuse* lni = nullptr) {
  m->act_planes.erase(out);   // (`out` is rewritten: planes registered for its previous contents are stale)
  const int rc = mlp_fused(m, m->gctx, st, rows, H, A, 256, m->P(p1 + ".weight"), m->P(p1 + ".bias"), m->P(p2 + ".weight"),
                           m->P(p2 + ".bias"), gamma, R, 256, out, 256, act, planes_too, f16x2, lnf, lni);
  if (ln_done) *ln_done = lnf && rc == DS2_OK;
  if (rc != DS2_ERR_UNSUPPORTED) return rc;
  if (lni) {   // the caller left the input un-normalised for the fused kernel's prologue and the kernel declined: that LayerNorm as its
               // own pass (operand planes for the first GEMM), then the two-GEMM form
    float* tn = m->alloc((size_t)rows * 256);
    ds2_model::ActPlanes op;
    op.hi = reinterpret_cast<unsigned short*>(m->alloc_bytes((size_t)rows * 512));
    op.lo = reinterpret_cast<unsigned short*>(m->alloc_bytes((size_t)rows * 512));
    op.ld = 256;
    if (!tn || !op.hi || !op.lo) { ds2_set_error("mlp2: workspace exhausted (LayerNorm fall-back)"); return DS2_ERR_STATE; }
    m->act_planes[tn] = op;
    TRY(launch_layernorm_split(A, 256, lni->w, lni->b, op.hi, op.lo, 256, rows, 256, lni->eps, DS2_ACT_NONE, st));
    A = tn;
  }
  TRY(linear(m, st, p1, rows, H, 256, A, 256, hbuf, H, act, nullptr, 0, 0, nullptr, true));
  return linear(m, st, p2, rows, 256, H, hbuf, H, out, 256, DS2_ACT_NONE, R, 256, 0, gamma);
}
// planes_out (bf16x3 mode only): the result is emitted as GEMM operand planes registered under key y; the fp32
// buffer y is not written, so every consumer of y must be a gemm()/linear().
static int layernorm(ds2_model* m, hipStream_t st, const std::string& p, const float* x, float* y, int rows, int C, float eps,
                     int act = DS2_ACT_NONE, bool planes_out = false, bool mx = false) {
  const float* w = m->P(p + ".weight");
  const float* b = m->P(p + ".bias");
  if (!w || !b) { ds2_set_error("missing parameter '%s'", p.c_str()); return DS2_ERR_STATE; }
  if (planes_out && ds2_split_mode()) {
    ds2_model::ActPlanes op;
    const int ld = round32i(C);
    op.hi = reinterpret_cast<unsigned short*>(m->alloc_bytes((size_t)rows * ld * 2));
    op.lo = reinterpret_cast<unsigned short*>(m->alloc_bytes((size_t)rows * ld * 2));
    op.ld = ld;
    op.fmt = mx ? DS2_PLANES_MX_A : DS2_PLANES_BF16;
    if (!op.hi || !op.lo) { ds2_set_error("workspace exhausted (layernorm planes)"); return DS2_ERR_STATE; }
    m->act_planes[y] = op;
    return launch_layernorm_split(x, C, w, b, op.hi, op.lo, ld, rows, C, eps, act, st, mx ? 1 : 0);
  }
  return launch_layernorm(x, C, w, b, y, C, rows, C, eps, act, st);
}

// ------------------------------------------------------------------------------------------------ lifetime
extern "C" int ds2_model_create(const ds2_config* cfg, ds2_model** out) {
  DS2_REQUIRE(cfg && out, "ds2_model_create: null argument");
  DS2_REQUIRE(cfg->image_size == 1024 && cfg->d_model == 256 && cfg->mem_dim == 64,
              "ds2_model_create: only image_size=1024, d_model=256, mem_dim=64 are supported");
  ds2_model* m = new ds2_model();
  m->cfg = *cfg;
  m->precision = g_ds2_default_precision;
  DS2_CHECK_HIP(hipGetDevice(&m->device));
  // per-block geometry: Hiera.__init__ loop (hieradet.py:236-267)
  int depth = 0;
  for (int s = 0; s < 4; ++s) { depth += cfg->stages[s]; m->stage_ends.push_back(depth - 1); }
  int dim = cfg->embed_dim, heads = cfg->num_heads, cur_stage = 1;
  for (int i = 0; i < depth; ++i) {
    int dim_out = dim, window = cfg->window_spec[cur_stage - 1];
    for (int g = 0; g < cfg->n_global_att_blocks; ++g)
      if (cfg->global_att_blocks[g] == i) window = 0;
    bool after_end = false;
    for (int s = 0; s < 4; ++s) after_end |= (m->stage_ends[s] == i - 1);
    if (after_end) { dim_out = dim * 2; heads *= 2; cur_stage += 1; }
    bool qpool = false;
    for (int s = 0; s < 3; ++s) qpool |= (m->stage_ends[s] + 1 == i);
    m->blocks.push_back({dim, dim_out, heads, window, qpool ? 2 : 0});
    dim = dim_out;
  }
  *out = m;
  return DS2_OK;
}

// A second execution context over the SAME weights: its own workspace arena, activation-plane table, GEMM scratch and
// arithmetic mode; parameters, derived constants and the bf16 planes of the weights are the parent's (which must outlive
// it).  For running one stage (e.g. the image encoder of the NEXT frames) on another stream concurrently with the parent.
extern "C" int ds2_model_create_view(ds2_model* parent, ds2_model** out) {
  DS2_REQUIRE(parent && out && parent->finalized && !parent->parent, "ds2_model_create_view: needs a finalized, non-view model");
  ds2_model* m = new ds2_model();
  m->cfg = parent->cfg;
  m->device = parent->device;
  m->precision = parent->precision;
  m->blocks = parent->blocks;
  m->stage_ends = parent->stage_ends;
  m->ma_fold_vo = parent->ma_fold_vo;
  m->parent = parent;
  m->gctx.share = &parent->gctx;
  m->finalized = true;
  *out = m;
  return DS2_OK;
}

extern "C" void ds2_model_destroy(ds2_model* m) {
  if (!m) return;
  ModelScope _dg(m);
  m->gctx.release();
  for (auto& kv : m->params)
    if (kv.second.ptr) (void)hipFree(kv.second.ptr);
  if (m->ws) (void)hipFree(m->ws);
  delete m;
}

extern "C" int ds2_model_set_precision(ds2_model* m, int32_t mode) {
  DS2_REQUIRE(m, "ds2_model_set_precision: null model");
  DS2_REQUIRE(mode == DS2_PREC_FP32 || mode == DS2_PREC_BF16X3 || mode == DS2_PREC_BF16X3K,
              "ds2_model_set_precision: mode must be 0 (fp32), 1 (bf16x3) or 2 (bf16x3k)");
  m->precision = mode;
  return DS2_OK;
}
extern "C" int ds2_model_get_precision(const ds2_model* m) { return m ? m->precision : -1; }

extern "C" int ds2_model_set_param(ds2_model* m, const char* name, const void* data, int64_t nbytes) {
  DS2_REQUIRE(m && name && data && nbytes > 0, "ds2_model_set_param: bad argument");
  ModelScope _dg(m);
  DS2_REQUIRE(!m->finalized && !m->parent, "ds2_model_set_param: model already finalized (or a view)");
  Blob b;
  b.bytes = (size_t)nbytes;
  DS2_CHECK_HIP(hipMalloc(&b.ptr, b.bytes));
  DS2_CHECK_HIP(hipMemcpy(b.ptr, data, b.bytes, hipMemcpyDefault));
  auto it = m->params.find(name);
  if (it != m->params.end() && it->second.ptr) (void)hipFree(it->second.ptr);
  m->params[name] = b;
  return DS2_OK;
}

static int expect(ds2_model* m, const std::string& name, size_t n_floats) {
  const size_t got = m->Pbytes(name);
  if (got == 0) { ds2_set_error("missing parameter '%s'", name.c_str()); return DS2_ERR_STATE; }
  if (got != n_floats * sizeof(float)) {
    ds2_set_error("parameter '%s' has %zu bytes, expected %zu", name.c_str(), got, n_floats * sizeof(float));
    return DS2_ERR_STATE;
  }
  return DS2_OK;
}

extern "C" int ds2_model_finalize(ds2_model* m, void* stream) {
  DS2_REQUIRE(m, "ds2_model_finalize: null model");
  ModelScope _dg(m);
  hipStream_t st = (hipStream_t)stream;
  const int C0 = m->cfg.embed_dim, D = 256;
  // ---- strict presence / size check of everything the stages read
  TRY(expect(m, "#pos_embed", (size_t)65536 * C0));
  TRY(expect(m, "#rope_cis", (size_t)TOK * 256));
  TRY(expect(m, "#vision_pos", (size_t)TOK * 256));
  TRY(expect(m, "#maskmem_pos", (size_t)TOK * 64));
  TRY(expect(m, "#dense_pe", (size_t)TOK * 256));
  TRY(expect(m, "#ptr_dim_t", 128));
  if (m->Pbytes("#ingest_lut") != 768 * sizeof(uint16_t)) { ds2_set_error("missing/bad '#ingest_lut'"); return DS2_ERR_STATE; }
  TRY(expect(m, "image_encoder.trunk.patch_embed.proj.weight", (size_t)C0 * 147));
  for (size_t i = 0; i < m->blocks.size(); ++i) {
    const BlockCfg& b = m->blocks[i];
    const std::string p = "image_encoder.trunk.blocks." + std::to_string(i);
    TRY(expect(m, p + ".norm1.weight", b.dim));
    TRY(expect(m, p + ".attn.qkv.weight", (size_t)3 * b.dim_out * b.dim));
    TRY(expect(m, p + ".attn.qkv.bias", (size_t)3 * b.dim_out));
    TRY(expect(m, p + ".attn.proj.weight", (size_t)b.dim_out * b.dim_out));
    TRY(expect(m, p + ".mlp.layers.0.weight", (size_t)4 * b.dim_out * b.dim_out));
    TRY(expect(m, p + ".mlp.layers.1.weight", (size_t)4 * b.dim_out * b.dim_out));
    if (b.dim != b.dim_out) TRY(expect(m, p + ".proj.weight", (size_t)b.dim_out * b.dim));
  }
  TRY(expect(m, "maskmem_tpos_enc", (size_t)m->cfg.num_maskmem * 64));
  TRY(expect(m, "obj_ptr_tpos_proj.weight", 64 * 256));
  TRY(expect(m, "memory_encoder.out_proj.weight", 64 * 256));
  TRY(expect(m, "sam_mask_decoder.output_upscaling.0.weight", 256 * 64 * 4));
  TRY(expect(m, "sam_mask_decoder.output_upscaling.3.weight", 64 * 32 * 4));

  float* w;
  // patch-embed weight [C,147] -> [C,148] (K padded to a multiple of 4)
  TRY(m->add_derived("@patch_w", (size_t)C0 * 148, &w));
  TRY(launch_pad_cols(m->P("image_encoder.trunk.patch_embed.proj.weight"), C0, 147, w, 148, st));
  // memory-attention self-attention: fused q|k|v projection [768,256]
  for (int l = 0; l < m->cfg.mem_attn_layers; ++l) {
    const std::string p = "memory_attention.layers." + std::to_string(l) + ".self_attn.";
    float *fw, *fb;
    TRY(m->add_derived("@ma_qkv_w." + std::to_string(l), 768 * 256, &fw));
    TRY(m->add_derived("@ma_qkv_b." + std::to_string(l), 768, &fb));
    const char* names[3] = {"q_proj", "k_proj", "v_proj"};
    // (fold: the value rows are out_proj o v_proj, constants.py fold_out_v - the attention output then IS the out_proj result)
    m->ma_fold_vo = m->Pbytes("#ma_self_vo_w." + std::to_string(l)) == 256 * 256 * 4 &&
                    m->Pbytes("#ma_cross_vo_w." + std::to_string(l)) == 256 * 64 * 4;
    for (int j = 0; j < 3; ++j) {
      TRY(expect(m, p + names[j] + ".weight", 256 * 256));
      TRY(expect(m, p + names[j] + ".bias", 256));
      const bool fold = j == 2 && m->ma_fold_vo;
      const float* wsrc = fold ? m->P("#ma_self_vo_w." + std::to_string(l)) : m->P(p + names[j] + ".weight");
      const float* bsrc = fold ? m->P("#ma_self_vo_b." + std::to_string(l)) : m->P(p + names[j] + ".bias");
      DS2_CHECK_HIP(hipMemcpyAsync(fw + j * 256 * 256, wsrc, 256 * 256 * 4, hipMemcpyDeviceToDevice, st));
      DS2_CHECK_HIP(hipMemcpyAsync(fb + j * 256, bsrc, 256 * 4, hipMemcpyDeviceToDevice, st));
    }
  }
  // ConvTranspose2d(2x2,s2) as GEMM: weight [Cin,Cout,2,2] -> [(dy,dx,cout), cin]; bias tiled 4x
  {
    float *w1, *b1, *w2, *b2;
    TRY(m->add_derived("@up1_w", 256 * 256, &w1));
    TRY(m->add_derived("@up1_b", 256, &b1));
    TRY(launch_permute4(m->P("sam_mask_decoder.output_upscaling.0.weight"), w1, 256, 64, 2, 2, 2, 3, 1, 0, st));
    TRY(m->add_derived("@up2_w", 128 * 64, &w2));
    TRY(m->add_derived("@up2_b", 128, &b2));
    TRY(launch_permute4(m->P("sam_mask_decoder.output_upscaling.3.weight"), w2, 64, 32, 2, 2, 2, 3, 1, 0, st));
    for (int j = 0; j < 4; ++j) {
      DS2_CHECK_HIP(hipMemcpyAsync(b1 + j * 64, m->P("sam_mask_decoder.output_upscaling.0.bias"), 64 * 4, hipMemcpyDeviceToDevice, st));
      DS2_CHECK_HIP(hipMemcpyAsync(b2 + j * 32, m->P("sam_mask_decoder.output_upscaling.3.bias"), 32 * 4, hipMemcpyDeviceToDevice, st));
    }
  }
  // two-way transformer: within a block keys + key_pe feeds k_proj of the token->image attention AND q_proj of the image->token
  // attention (transformer.py:196-197,211-212): the two [128,256] weights stacked to one [256,256] GEMM operand
  for (int l = 0; l < 2; ++l) {
    const std::string p = "sam_mask_decoder.transformer.layers." + std::to_string(l);
    const std::string a = p + ".cross_attn_token_to_image.k_proj", b = p + ".cross_attn_image_to_token.q_proj";
    if (m->Pbytes(a + ".weight") != 128 * 256 * 4 || m->Pbytes(b + ".weight") != 128 * 256 * 4 || m->Pbytes(a + ".bias") != 128 * 4 ||
        m->Pbytes(b + ".bias") != 128 * 4)
      continue;   // (not the SAM 2 shapes: sam_attention falls back to the separate projections)
    float *fw, *fb;
    TRY(m->add_derived("@sam_kq_w." + std::to_string(l), 256 * 256, &fw));
    TRY(m->add_derived("@sam_kq_b." + std::to_string(l), 256, &fb));
    DS2_CHECK_HIP(hipMemcpyAsync(fw, m->P(a + ".weight"), 128 * 256 * 4, hipMemcpyDeviceToDevice, st));
    DS2_CHECK_HIP(hipMemcpyAsync(fw + 128 * 256, m->P(b + ".weight"), 128 * 256 * 4, hipMemcpyDeviceToDevice, st));
    DS2_CHECK_HIP(hipMemcpyAsync(fb, m->P(a + ".bias"), 128 * 4, hipMemcpyDeviceToDevice, st));
    DS2_CHECK_HIP(hipMemcpyAsync(fb + 128, m->P(b + ".bias"), 128 * 4, hipMemcpyDeviceToDevice, st));
  }
  // ... and the token side: projections that read the same tokens stacked row-wise into one few-row Linear each
  // (k_skinny_linear: a row's result depends on K only, so stacking is bit-identical)
  {
    const std::string t = "sam_mask_decoder.transformer.";
    auto stack = [&](const std::string& name, std::initializer_list<std::string> parts) -> int {
      size_t rows = 0;
      for (const std::string& q : parts) {
        const size_t wb = m->Pbytes(q + ".weight"), bb = m->Pbytes(q + ".bias");
        if (!wb || wb % (256 * 4) || bb != wb / 256) return DS2_OK;   // other shapes: the separate projections are used
        rows += bb / 4;
      }
      float *fw, *fb;
      TRY(m->add_derived(name + "_w", rows * 256, &fw));
      TRY(m->add_derived(name + "_b", rows, &fb));
      size_t r = 0;
      for (const std::string& q : parts) {
        const size_t n = m->Pbytes(q + ".bias") / 4;
        DS2_CHECK_HIP(hipMemcpyAsync(fw + r * 256, m->P(q + ".weight"), n * 256 * 4, hipMemcpyDeviceToDevice, st));
        DS2_CHECK_HIP(hipMemcpyAsync(fb + r, m->P(q + ".bias"), n * 4, hipMemcpyDeviceToDevice, st));
        r += n;
      }
      return DS2_OK;
    };
    const std::string l0 = t + "layers.0.", l1 = t + "layers.1.";
    TRY(stack("@sam_s0_qkv", {l0 + "self_attn.q_proj", l0 + "self_attn.k_proj", l0 + "self_attn.v_proj"}));
    TRY(stack("@sam_tq.0", {l0 + "cross_attn_image_to_token.k_proj", l1 + "self_attn.q_proj", l1 + "self_attn.k_proj"}));
    TRY(stack("@sam_tv.0", {l0 + "cross_attn_image_to_token.v_proj", l1 + "self_attn.v_proj"}));
    TRY(stack("@sam_tq.1", {l1 + "cross_attn_image_to_token.k_proj", t + "final_attn_token_to_image.q_proj"}));
  }
  // mask-downsampler 3x3 convs as im2col GEMMs: [Cout,Cin,3,3] -> [Cout,(ky,kx,cin)]
  TRY(m->add_derived("@mds6_w", 64 * 144, &w));
  TRY(launch_permute4(m->P("memory_encoder.mask_downsampler.encoder.6.weight"), w, 64, 16, 3, 3, 0, 2, 3, 1, st));
  TRY(m->add_derived("@mds9_w", 256 * 576, &w));
  TRY(launch_permute4(m->P("memory_encoder.mask_downsampler.encoder.9.weight"), w, 256, 64, 3, 3, 0, 2, 3, 1, st));
  // CXBlock depth-wise weights [C,1,7,7] -> [49][C]
  for (int l = 0; l < 2; ++l) {
    TRY(m->add_derived("@dw_w." + std::to_string(l), 49 * 256, &w));
    TRY(launch_permute4(m->P("memory_encoder.fuser.layers." + std::to_string(l) + ".dwconv.weight"), w, 256, 1, 7, 7, 2, 3, 0, 1, st));
  }
  // decoder output tokens [obj_score, iou, mask0..3] and point-label embeddings [4,256]
  TRY(m->add_derived("@out_tokens6", 6 * 256, &w));
  DS2_CHECK_HIP(hipMemcpyAsync(w, m->P("sam_mask_decoder.obj_score_token.weight"), 256 * 4, hipMemcpyDeviceToDevice, st));
  DS2_CHECK_HIP(hipMemcpyAsync(w + 256, m->P("sam_mask_decoder.iou_token.weight"), 256 * 4, hipMemcpyDeviceToDevice, st));
  DS2_CHECK_HIP(hipMemcpyAsync(w + 512, m->P("sam_mask_decoder.mask_tokens.weight"), 4 * 256 * 4, hipMemcpyDeviceToDevice, st));
  TRY(m->add_derived("@point_emb4", 4 * 256, &w));
  for (int j = 0; j < 4; ++j)
    DS2_CHECK_HIP(hipMemcpyAsync(w + j * 256, m->P("sam_prompt_encoder.point_embeddings." + std::to_string(j) + ".weight"),
                                 256 * 4, hipMemcpyDeviceToDevice, st));
  // dense-prompt vectors: no_mask_embed, and no_mask_embed + no_mem_embed (init-cond frames)
  TRY(m->add_derived("@dense_vec", 256, &w));
  DS2_CHECK_HIP(hipMemcpyAsync(w, m->P("sam_prompt_encoder.no_mask_embed.weight"), 256 * 4, hipMemcpyDeviceToDevice, st));
  CHECK_PARAMS();
  DS2_CHECK_HIP(hipStreamSynchronize(st));
  m->finalized = true;
  (void)D;
  return DS2_OK;
}

// ------------------------------------------------------------------------------------------------ A3
// OpenCV INTER_LINEAR tap/weight tables for 8-bit resize (resize.cpp): f = (float)((d + 0.5) * scale - 0.5) with the
// scale in double, tap = floor(f), weights cvRound(w * 2048) saturated to int16.  The x pass zeroes the fraction when
// the tap is clamped; the y pass keeps it and only the row indices are clipped (in the kernel).
static void resize_tables(int dst, int src, bool clamp_weights, int* ofs, int* w0, int* w1) {
  const double scale = (double)src / (double)dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_weights) {
      if (s < 0) { f = 0.f; s = 0; }
      if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    long a1 = lrintf(f * 2048.f), a0 = lrintf((1.f - f) * 2048.f);   // round-half-even = cvRound
    a1 = a1 < -32768 ? -32768 : (a1 > 32767 ? 32767 : a1);
    a0 = a0 < -32768 ? -32768 : (a0 > 32767 ? 32767 : a0);
    ofs[d] = s; w0[d] = (int)a0; w1[d] = (int)a1;
  }
}

extern "C" int ds2_ingest_frames(ds2_model* m, const uint8_t* rgb_u8, int32_t n, int32_t height, int32_t width,
                                 uint16_t* frames_f16, void* stream) {
  DS2_REQUIRE(m && m->finalized && rgb_u8 && frames_f16 && n > 0 && height > 0 && width > 0, "ds2_ingest_frames: bad argument");
  ModelScope _dg(m);
  const int S = m->cfg.image_size;
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* lut = reinterpret_cast<const uint16_t*>(m->P("#ingest_lut"));
  if (height == S && width == S) return launch_ingest_u8(rgb_u8, lut, frames_f16, n, S, st);   // cv::resize early-out: copy
  const std::string key = "@resize_tab." + std::to_string(height) + "x" + std::to_string(width);
  const int* tab = m->Pbytes(key) ? reinterpret_cast<const int*>(m->P(key)) : nullptr;   // Pbytes: probe without flagging
  if (!tab) {   // first frame of this resolution: build the six tables once, keep them on the device
    std::vector<int> h((size_t)6 * S);
    resize_tables(S, width, true, h.data(), h.data() + S, h.data() + 2 * S);
    resize_tables(S, height, false, h.data() + 3 * S, h.data() + 4 * S, h.data() + 5 * S);
    float* d = nullptr;
    TRY(m->add_derived(key, (size_t)6 * S, &d));
    DS2_CHECK_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, st));
    DS2_CHECK_HIP(hipStreamSynchronize(st));   // h is a local buffer
    tab = reinterpret_cast<const int*>(d);
  }
  return launch_ingest_resize_u8(rgb_u8, tab, lut, frames_f16, n, height, width, S, st);
}

// ------------------------------------------------------------------------------------------------ A4 + A5
static int image_encoder_impl(ds2_model* m, const void* frames, bool frames_f32, int32_t n, float* fpn0, float* fpn1, float* fpn2,
                              void* stream);
extern "C" int ds2_image_encoder_batch(ds2_model* m, const uint16_t* frames_f16, int32_t n, float* fpn0, float* fpn1, float* fpn2,
                                       void* stream) {
  return image_encoder_impl(m, frames_f16, false, n, fpn0, fpn1, fpn2, stream);
}
extern "C" int ds2_image_encoder_f32(ds2_model* m, const float* frames_f32, int32_t n, float* fpn0, float* fpn1, float* fpn2,
                                     void* stream) {
  return image_encoder_impl(m, frames_f32, true, n, fpn0, fpn1, fpn2, stream);
}
static int image_encoder_impl(ds2_model* m, const void* frames, bool frames_f32, int32_t n, float* fpn0, float* fpn1, float* fpn2,
                              void* stream) {
  const uint16_t* frames_f16 = reinterpret_cast<const uint16_t*>(frames);
  DS2_REQUIRE(m && m->finalized && frames_f16 && fpn0 && fpn1 && fpn2 && n >= 1 && n <= 16, "ds2_image_encoder_batch: bad argument");
  ModelScope _dg(m);
  hipStream_t st = (hipStream_t)stream;
  ProfScope _ps("stage.image_encoder", st);
  const int C0 = m->cfg.embed_dim;
  // workspace bound: every block output kept (sum <= depth * 65536*C0 floats is a loose bound) + temporaries
  size_t need = (size_t)n * 65536 * 148 * 4;
  {
    int side = 256;
    size_t outs = 0, tmp_max = 0;
    for (const BlockCfg& b : m->blocks) {
      const size_t hw = (size_t)n * side * side, hwq = b.q_stride ? hw / 4 : hw;
      outs += hwq * b.dim_out * 4 + 256;
      // fp32 temporaries + (bf16x3 mode) their operand planes, which take the same number of bytes
      size_t tmp = 2 * (hw * (b.dim + 32) + hw * b.dim_out * 2 + hw * 3 * b.dim_out + hwq * (b.dim_out + 32) * 3 + hwq * 4 * b.dim_out) * 4 + 65536;
      if (b.window == 0) tmp += hw * b.heads * (size_t)(((b.dim_out / b.heads + 31) / 32 * 32) * 4 + 64 + ((b.dim_out / b.heads + 15) / 16 * 16) * 6) + 4096;   // attention_hg tile images
      if (tmp > tmp_max) tmp_max = tmp;
      if (b.q_stride) side /= 2;
    }
    need += outs + tmp_max + (size_t)n * (65536 + 16384 + 4096 * 2 + 1024) * 256 * 4 + (1u << 20);
  }
  TRY(m->require(need, st));

  // PatchEmbed (backbones/utils.py:93-96) + pos embed (hieradet.py:273-281), fused as GEMM epilogue
  ALLOC(col, (size_t)n * 65536 * 148);
  for (int i = 0; i < n; ++i) {
    if (frames_f32)
      TRY(launch_im2col_patch_f32(reinterpret_cast<const float*>(frames) + (size_t)i * 3 * 1024 * 1024, col + (size_t)i * 65536 * 148, 1024, st));
    else
      TRY(launch_im2col_patch(frames_f16 + (size_t)i * 3 * 1024 * 1024, col + (size_t)i * 65536 * 148, 1024, st));
  }
  ALLOC(x0, (size_t)n * 65536 * C0);
  TRY(gemm(st, n * 65536, C0, 148, col, 148, m->P("@patch_w"), 148, m->P("image_encoder.trunk.patch_embed.proj.bias"), x0, C0,
           DS2_ACT_NONE, m->P("#pos_embed"), C0, 65536, nullptr, true, m));
  const float* x = x0;
  int side = 256;
  const float* stage_out[4] = {nullptr, nullptr, nullptr, nullptr};
  int stage_dim[4] = {0, 0, 0, 0}, stage_side[4] = {0, 0, 0, 0};
  int n_stage = 0;
  for (size_t i = 0; i < m->blocks.size(); ++i) {
    const BlockCfg& b = m->blocks[i];
    const std::string p = "image_encoder.trunk.blocks." + std::to_string(i);
    const int hw1 = side * side, hw = n * hw1;                                   // tokens per image / in the batch
    const int side_q = b.q_stride ? side / 2 : side, hwq1 = side_q * side_q, hwq = n * hwq1;
    ALLOC(xn, (size_t)hwq * b.dim_out);           // block output survives the temporaries below
    const size_t mark = m->ws_top;
    ALLOC(t, (size_t)hw * b.dim);
    // (round 6) operand planes go out in the format their consumer multiplies in: "MX" planes where the Linear layer takes the
    // two-MFMA-equivalent form (hiera_l stages 3 / 4), bf16 hi / lo everywhere else
    const bool mx_qkv = linear_mx(m, p + ".attn.qkv", hw, 3 * b.dim_out, b.dim, DS2_ACT_NONE, false, false) &&
                        (b.dim == b.dim_out || linear_mx(m, p + ".proj", hw, b.dim_out, b.dim, DS2_ACT_NONE, false, false));
    const bool mx_proj = linear_mx(m, p + ".attn.proj", hwq, b.dim_out, b.dim_out, DS2_ACT_NONE, true, false);
    const bool mx_mlp = linear_mx(m, p + ".mlp.layers.0", hwq, 4 * b.dim_out, b.dim_out, DS2_ACT_GELU, false, true) &&
                        linear_mx(m, p + ".mlp.layers.1", hwq, b.dim_out, 4 * b.dim_out, DS2_ACT_NONE, true, false);   // (both or neither)
    TRY(layernorm(m, st, p + ".norm1", x, t, hw, b.dim, 1e-6f, DS2_ACT_NONE, true, mx_qkv));   // consumers: proj / qkv GEMMs
    const float* sc = x;
    if (b.dim != b.dim_out) {                      // hieradet.py:141-142
      ALLOC(scf, (size_t)hw * b.dim_out);
      TRY(linear(m, st, p + ".proj", hw, b.dim_out, b.dim, t, b.dim, scf, b.dim_out));
      if (b.q_stride) {
        ALLOC(scp, (size_t)hwq * b.dim_out);
        for (int i = 0; i < n; ++i)
          TRY(launch_maxpool2x2(scf + (size_t)i * hw1 * b.dim_out, b.dim_out, scp + (size_t)i * hwq1 * b.dim_out, b.dim_out, side, side,
                                b.dim_out, st));
        sc = scp;
      } else {
        sc = scf;
      }
    }
    ALLOC(qkv, (size_t)hw * 3 * b.dim_out);
    TRY(linear(m, st, p + ".attn.qkv", hw, 3 * b.dim_out, b.dim, t, b.dim, qkv, 3 * b.dim_out));
    const float* q = qkv;
    int ldq = 3 * b.dim_out;
    if (b.q_stride) {                              // q pooling (hieradet.py:65-68); even windows => plain 2x2 pooling
      ALLOC(qp, (size_t)hwq * b.dim_out);
      for (int i = 0; i < n; ++i)
        TRY(launch_maxpool2x2(qkv + (size_t)i * hw1 * 3 * b.dim_out, 3 * b.dim_out, qp + (size_t)i * hwq1 * b.dim_out, b.dim_out, side,
                              side, b.dim_out, st));
      q = qp;
      ldq = b.dim_out;
    }
    ALLOC(a, (size_t)hwq * b.dim_out);
    AttnArgs aa{};
    aa.q = q; aa.k = qkv + b.dim_out; aa.v = qkv + 2 * b.dim_out; aa.o = a;
    aa.ldq = ldq; aa.ldk = aa.ldv = 3 * b.dim_out; aa.ldo = b.dim_out;
    aa.heads = b.heads; aa.D = aa.DV = b.dim_out / b.heads;
    aa.scale = 1.0f / sqrtf((float)aa.D);
    if (b.window == 0) {
      aa.batch = n; aa.Lq = hwq1; aa.Lk = hw1; aa.win_q = aa.win_k = 0;
    } else {
      const int nw = cdiv(side, b.window);
      aa.win_k = b.window; aa.win_q = b.q_stride ? b.window / 2 : b.window;
      aa.batch = n * nw * nw; aa.nwx = nw; aa.wins = nw * nw;
      aa.Lq = aa.win_q * aa.win_q; aa.Lk = aa.win_k * aa.win_k;
      aa.Hq = aa.Wq = side_q; aa.Hk = aa.Wk = side;
      const float* qb = m->P(p + ".attn.qkv.bias");
      aa.k_pad = qb ? qb + b.dim_out : nullptr;
      aa.v_pad = qb ? qb + 2 * b.dim_out : nullptr;
    }
    if (ds2_split_mode()) {   // attention output goes straight to the proj GEMM: emit planes
      ds2_model::ActPlanes ap;
      TRY(new_act_planes(m, a, hwq, b.dim_out, &ap, st));
      aa.o_hi = ap.hi; aa.o_lo = ap.lo; aa.ldop = ap.ld;
      if (mx_proj) { aa.o_mx = 1; m->act_planes[a].fmt = DS2_PLANES_MX_A; }
    }
    {
      ProfScope _pa("kernel.hiera_attention", st);
      // global attention in the split modes: K / V pre-split once per (image, head), attention_hg.hip
      if (ds2_split_mode() && attention_hg_supported(aa)) {
        void* kimg = m->alloc_bytes(attention_hg_k_bytes(aa));
        void* vimg = m->alloc_bytes(attention_hg_vt_bytes(aa));
        if (!kimg || !vimg) { ds2_set_error("image encoder: workspace exhausted (global attention planes)"); return DS2_ERR_STATE; }
        TRY(launch_attention_hg(aa, kimg, vimg, st));
      } else {
        TRY(launch_attention(aa, st));
      }
    }
    TRY(linear(m, st, p + ".attn.proj", hwq, b.dim_out, b.dim_out, a, b.dim_out, xn, b.dim_out, DS2_ACT_NONE, sc, b.dim_out));
    ALLOC(t2, (size_t)hwq * b.dim_out);
    TRY(layernorm(m, st, p + ".norm2", xn, t2, hwq, b.dim_out, 1e-6f, DS2_ACT_NONE, true, mx_mlp));
    ALLOC(h, (size_t)hwq * 4 * b.dim_out);
    TRY(linear(m, st, p + ".mlp.layers.0", hwq, 4 * b.dim_out, b.dim_out, t2, b.dim_out, h, 4 * b.dim_out, DS2_ACT_GELU,
               nullptr, 0, 0, nullptr, true, mx_mlp));   // hidden activations only feed mlp.layers.1: planes only
    TRY(linear(m, st, p + ".mlp.layers.1", hwq, b.dim_out, 4 * b.dim_out, h, 4 * b.dim_out, xn, b.dim_out, DS2_ACT_NONE, xn, b.dim_out));
    m->release(mark);
    x = xn;
    side = side_q;
    for (int s = 0; s < 4; ++s)
      if (m->stage_ends[s] == (int)i) { stage_out[n_stage] = xn; stage_dim[n_stage] = b.dim_out; stage_side[n_stage] = side; ++n_stage; }
  }
  DS2_REQUIRE(n_stage == 4, "image encoder: expected 4 stage outputs, got %d", n_stage);
  // FpnNeck (image_encoder.py:101-134): convs[n-i] on xs[i]; top-down nearest only into level 2; scalp=1
  ALLOC(lat3, (size_t)n * 1024 * 256);
  ALLOC(lat2, (size_t)n * 4096 * 256);
  ALLOC(lat1, (size_t)n * 16384 * 256);
  ALLOC(lat0, (size_t)n * 65536 * 256);
  TRY(linear(m, st, "image_encoder.neck.convs.0.conv", n * 1024, 256, stage_dim[3], stage_out[3], stage_dim[3], lat3, 256));
  TRY(linear(m, st, "image_encoder.neck.convs.1.conv", n * 4096, 256, stage_dim[2], stage_out[2], stage_dim[2], lat2, 256));
  TRY(linear(m, st, "image_encoder.neck.convs.2.conv", n * 16384, 256, stage_dim[1], stage_out[1], stage_dim[1], lat1, 256));
  TRY(linear(m, st, "image_encoder.neck.convs.3.conv", n * 65536, 256, stage_dim[0], stage_out[0], stage_dim[0], lat0, 256));
  for (int i = 0; i < n; ++i)
    TRY(launch_up2_add(lat2 + (size_t)i * 4096 * 256, lat3 + (size_t)i * 1024 * 256, fpn2 + (size_t)i * 4096 * 256, 64, 64, 256, st));
  // conv_s0 / conv_s1 (sam2_base.py:455-460)
  TRY(linear(m, st, "sam_mask_decoder.conv_s1", n * 16384, 64, 256, lat1, 256, fpn1, 64));
  TRY(linear(m, st, "sam_mask_decoder.conv_s0", n * 65536, 32, 256, lat0, 256, fpn0, 32));
  CHECK_PARAMS();
  (void)stage_side;
  return DS2_OK;
}

extern "C" int ds2_image_encoder(ds2_model* m, const uint16_t* frame_f16, float* fpn0, float* fpn1, float* fpn2, void* stream) {
  return ds2_image_encoder_batch(m, frame_f16, 1, fpn0, fpn1, fpn2, stream);
}

// ------------------------------------------------------------------------------------------------ A11
namespace {
struct BankSrc {   // the entry tables of ds2_bank_assemble (HOST arrays of device pointers)
  int n_mem; const void* const* feats; const int32_t* tpos_row;
  int n_ptr; const float* const* ptrs; const float* ptr_pos;
};
int bank_check(ds2_model* m, const BankSrc& s, const char* who) {
  DS2_REQUIRE(s.n_mem >= 0 && s.n_ptr >= 0 && (s.n_mem == 0 || (s.feats && s.tpos_row)) && (s.n_ptr == 0 || (s.ptrs && s.ptr_pos)),
              "%s: bad entry tables (n_mem=%d, n_ptr=%d)", who, s.n_mem, s.n_ptr);
  for (int e = 0; e < s.n_mem; ++e) {
    DS2_REQUIRE(s.feats[e], "%s: null feature pointer", who);
    DS2_REQUIRE(s.tpos_row[e] >= 0 && s.tpos_row[e] < m->cfg.num_maskmem, "%s: bad tpos_row", who);
  }
  for (int i = 0; i < s.n_ptr; ++i) DS2_REQUIRE(s.ptrs[i], "%s: null pointer entry", who);
  return DS2_OK;
}
// the entry tables travel in the kernel arguments, DS2_MAX_*_ENTRIES at a time (no limit on the bank size); fm(a) / fp(a) launch for one batch
template <class FM, class FP>
int bank_for_each(ds2_model* m, int B, const BankSrc& s, float* memory, float* memory_pos, FM fm, FP fp) {
  BankArgs a{};
  a.B = B; a.tokens = TOK;
  a.Nk = s.n_mem * TOK + 4 * s.n_ptr; a.n_mem_total = s.n_mem;
  a.maskmem_pos = m->P("#maskmem_pos");
  a.tpos_enc = m->P("maskmem_tpos_enc");
  a.tpos_w = m->P("obj_ptr_tpos_proj.weight");
  a.tpos_b = m->P("obj_ptr_tpos_proj.bias");
  a.mem = memory; a.mem_pos = memory_pos;
  CHECK_PARAMS();
  for (int e0 = 0; e0 < s.n_mem; e0 += DS2_MAX_MEM_ENTRIES) {
    a.e0 = e0; a.n_mem = s.n_mem - e0 < DS2_MAX_MEM_ENTRIES ? s.n_mem - e0 : DS2_MAX_MEM_ENTRIES; a.n_ptr = 0;
    for (int e = 0; e < a.n_mem; ++e) {
      a.feats[e] = reinterpret_cast<const uint16_t*>(s.feats[e0 + e]);
      a.tpos_row[e] = s.tpos_row[e0 + e];
    }
    TRY(fm(a));
  }
  for (int p0 = 0; p0 < s.n_ptr; p0 += DS2_MAX_PTR_ENTRIES) {
    a.p0 = p0; a.n_ptr = s.n_ptr - p0 < DS2_MAX_PTR_ENTRIES ? s.n_ptr - p0 : DS2_MAX_PTR_ENTRIES; a.n_mem = 0;
    for (int i = 0; i < a.n_ptr; ++i) { a.ptrs[i] = s.ptrs[p0 + i]; a.ptr_pos[i] = s.ptr_pos[p0 + i]; }
    TRY(fp(a));
  }
  return DS2_OK;
}
int bank_to_fp32(ds2_model* m, int B, const BankSrc& s, float* memory, float* memory_pos, hipStream_t st) {
  return bank_for_each(m, B, s, memory, memory_pos, [&](const BankArgs& a) { return launch_bank_assemble(a, st); },
                       [&](const BankArgs& a) { return launch_bank_ptr(a, m->P("#ptr_dim_t"), st); });
}
}  // namespace
extern "C" int ds2_bank_assemble(ds2_model* m, int32_t B, int32_t n_mem, const void* const* feats, const int32_t* tpos_row,
                                 int32_t n_ptr, const float* const* ptrs, const float* ptr_pos, float* memory,
                                 float* memory_pos, void* stream) {
  DS2_REQUIRE(m && m->finalized && B > 0 && memory && memory_pos, "ds2_bank_assemble: bad argument");
  ModelScope _dg(m);
  const BankSrc s{n_mem, feats, tpos_row, n_ptr, ptrs, ptr_pos};
  TRY(bank_check(m, s, "ds2_bank_assemble"));
  return bank_to_fp32(m, B, s, memory, memory_pos, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ A12
static int memory_attention_impl(ds2_model* m, int32_t B, const float* curr, bool curr_shared, const float* curr_pos,
                                 bool pos_shared, const float* memory, const float* memory_pos, int32_t Nk, int32_t n_ptr_tok,
                                 float* out, void* stream, const BankSrc* bank = nullptr);
extern "C" int ds2_memory_attention(ds2_model* m, int32_t B, const float* curr, const float* memory, const float* memory_pos,
                                    int32_t Nk, int32_t n_ptr_tok, float* out, void* stream) {
  return memory_attention_impl(m, B, curr, true, nullptr, true, memory, memory_pos, Nk, n_ptr_tok, out, stream);
}
extern "C" int ds2_memory_attention_ex(ds2_model* m, int32_t B, const float* curr, int32_t curr_shared, const float* curr_pos,
                                       int32_t pos_shared, const float* memory, const float* memory_pos, int32_t Nk,
                                       int32_t n_ptr_tok, float* out, void* stream) {
  return memory_attention_impl(m, B, curr, curr_shared != 0, curr_pos, pos_shared != 0, memory, memory_pos, Nk, n_ptr_tok, out, stream);
}
// bank != nullptr (ds2_bank_memory_attention): memory / memory_pos are NOT given - the bank's entries are turned into the cross-attention's
// operands directly (kin planes, V^T tiles) when the assembly attention runs, else assembled as fp32 tensors in the workspace first
// the queries of the memory cross-attention of one layer as the assembly attention's Q fragments, by the fused kernel (fused != 0:
// gemm_qproj.hip) or by the three kernels it replaces (LayerNorm pass -> GEMM -> query pass): unit test of the former against the latter
extern "C" int ds2_op_query_fragments(ds2_model* m, int32_t layer, const float* x, int32_t rows, int32_t fused, void* qfrag, void* stream) {
  DS2_REQUIRE(m && m->finalized && x && qfrag && rows > 0 && rows % 64 == 0 && layer >= 0 && layer < m->cfg.mem_attn_layers,
              "ds2_op_query_fragments: bad argument");
  ModelScope _dg(m);
  hipStream_t st = (hipStream_t)stream;
  const std::string p = "memory_attention.layers." + std::to_string(layer);
  const float* cis = m->P("#rope_cis");
  const float* qw = m->P(p + ".cross_attn_image.q_proj.weight");
  const float sc = 1.0f / 16.0f;
  CHECK_PARAMS();
  if (fused) {
    DS2_REQUIRE(ds2_split_mode() && qproj_x4a_supported(rows, 256, 256), "ds2_op_query_fragments: the fused kernel needs a split mode");
    GemmPlanes qwp;
    TRY(weight_planes(m->gctx, qw, 256, 256, &qwp, st));
    return launch_qproj_x4a(x, 256, rows, m->P(p + ".norm2.weight"), m->P(p + ".norm2.bias"), 1e-5f, qwp.hi, qwp.lo, qwp.ld,
                            m->P(p + ".cross_attn_image.q_proj.bias"), cis, TOK, sc, qfrag, 1, st);
  }
  TRY(m->require((size_t)rows * 256 * 4 * 4 + (8u << 20), st));
  ALLOC(t, (size_t)rows * 256);
  ALLOC(q, (size_t)rows * 256);
  TRY(layernorm(m, st, p + ".norm2", x, t, rows, 256, 1e-5f, DS2_ACT_NONE, true));
  TRY(linear(m, st, p + ".cross_attn_image.q_proj", rows, 256, 256, t, 256, q, 256));
  m->act_planes.erase(t);
  return launch_x4a_qprep(q, 256, 1, rows, false, sc, cis, TOK, qfrag, st);
}

extern "C" int ds2_bank_memory_attention(ds2_model* m, int32_t B, const float* curr, int32_t n_mem, const void* const* feats,
                                         const int32_t* tpos_row, int32_t n_ptr, const float* const* ptrs, const float* ptr_pos,
                                         float* out, void* stream) {
  DS2_REQUIRE(m && m->finalized && B > 0 && curr && out, "ds2_bank_memory_attention: bad argument");
  const BankSrc s{n_mem, feats, tpos_row, n_ptr, ptrs, ptr_pos};
  TRY(bank_check(m, s, "ds2_bank_memory_attention"));
  return memory_attention_impl(m, B, curr, true, nullptr, true, nullptr, nullptr, n_mem * TOK + 4 * n_ptr, 4 * n_ptr, out, stream, &s);
}
static int memory_attention_impl(ds2_model* m, int32_t B, const float* curr, bool curr_shared, const float* curr_pos,
                                 bool pos_shared, const float* memory, const float* memory_pos, int32_t Nk, int32_t n_ptr_tok,
                                 float* out, void* stream, const BankSrc* bank) {
  DS2_REQUIRE(m && m->finalized && B > 0 && curr && (bank || (memory && memory_pos)) && out && Nk > 0 && n_ptr_tok >= 0 && n_ptr_tok <= Nk,
              "ds2_memory_attention: bad argument");
  ModelScope _dg(m);
  // the layer-0 self-attention is shared by the B objects only when they all see the same tokens AND positions
  const bool shared0 = curr_shared && (curr_pos == nullptr || pos_shared);
  DS2_REQUIRE((Nk - n_ptr_tok) % TOK == 0, "ds2_memory_attention: Nk - num_obj_ptr_tokens must be a multiple of 4096");
  // a bank of object pointers alone (no memory frame) never occurs in propagate_in_video (the first tracked frame attends its
  // conditioning frame, sam2_base.py:526-585), and the single-plane fp16 keys of mode bf16x3k are written by the K = 64 streaming kernel,
  // which wants >= 4096 key rows: say so instead of failing inside the GEMM dispatch
  DS2_REQUIRE(Nk - n_ptr_tok >= TOK, "ds2_memory_attention: the bank must hold at least one memory frame (Nk - num_obj_ptr_tokens >= 4096, got %d)",
              Nk - n_ptr_tok);
  hipStream_t st = (hipStream_t)stream;
  ProfScope _ps("stage.memory_attention", st);
  const int rows = B * TOK, F = m->cfg.mem_attn_ffn;
  const bool split = ds2_split_mode();
  const bool klo_planes = ds2_precision() != DS2_PREC_BF16X3K;   // keys of the attention scores carry a lo plane
  const bool k_f16 = !klo_planes && DS2_ATTN_K_F16;              // bf16x3k: the single operand planes are fp16
  const int nt_c = (Nk + 31) / 32, nt_s = TOK / 32;
  const size_t split_bytes = split ? ((size_t)B * Nk * 256 * 4 + (size_t)B * nt_c * 8192 + (size_t)rows * 256 * 4 +
                                      (size_t)4 * B * nt_s * 8192 + (1u << 20)) : 0;
  const size_t plane_bytes = split ? ((size_t)rows * (256 * 5 + 64 + F) + (size_t)B * Nk * 64) * 4 + (8u << 20) : 0;
  // scratch of the key-split attention (few workgroups: attention_w8.hip): nsplit * batch <= 8 query sets of [4096, 256 + 2]
  const size_t ksplit_bytes = split ? (size_t)8 * TOK * (256 + 2) * sizeof(float) : 0;
  // the assembly cross-attention (attention_x4a.hip): its own V^T tile order, Q fragments and unnormalised partial rows
  const bool x4a = split && k_f16 && attention_x4a_enabled() && attention_x4a_supported(B, TOK, Nk, 64, true);
  const size_t x4a_ws_bytes = x4a ? attention_x4a_ws_bytes(B, TOK, Nk) : 0;
  const size_t x4a_bytes = x4a ? x4a_ws_bytes + (size_t)B * nt_c * 4096 + 4096 : 0;
  const bool bank_direct = bank && x4a;       // entries -> kin planes + V^T tiles, no fp32 memory / memory_pos
  const size_t bank_fp32_bytes = (bank && !bank_direct) ? (size_t)2 * B * Nk * 64 * 4 + 512 : 0;
  const size_t need = ((size_t)rows * (256 * 5 + 768 + 64 + F) + (size_t)B * Nk * (64 + 256) + (size_t)TOK * 256 * 4) * 4 + split_bytes + plane_bytes + ksplit_bytes + x4a_bytes + (split ? (size_t)rows * 2048 + 4096 : 0) + mlp256_part_bytes(rows, F) + bank_fp32_bytes + (4u << 20);
  TRY(m->require(need, st));
  if (bank && !bank_direct) {   // every other mode: the fp32 tensors as ds2_bank_assemble builds them, in the workspace
    float* mf = reinterpret_cast<float*>(m->alloc_bytes((size_t)B * Nk * 64 * 4));
    float* pf = reinterpret_cast<float*>(m->alloc_bytes((size_t)B * Nk * 64 * 4));
    if (!mf || !pf) { ds2_set_error("memory_attention: workspace exhausted"); return DS2_ERR_STATE; }
    TRY(bank_to_fp32(m, B, *bank, mf, pf, st));
    memory = mf; memory_pos = pf;
  }
  const float* cis = m->P("#rope_cis");
  ALLOC(x, (size_t)rows * 256);
  ALLOC(x1, (size_t)TOK * 256);
  ALLOC(t, (size_t)rows * 256);
  ALLOC(qkv, (size_t)rows * 768);
  ALLOC(a, (size_t)rows * 256);
  ALLOC(q, (size_t)rows * 256);
  ALLOC(a64, (size_t)rows * 64);
  ALLOC(kin, (size_t)B * Nk * 64);
  ALLOC(K, (size_t)B * Nk * 256);
  ALLOC(h, (size_t)rows * F);
  // bf16x3 mode: attention operands are split into bf16 planes once by their producers (attention_split.hip)
  void *khi = nullptr, *klo = nullptr, *vt_c = nullptr, *khi_s = nullptr, *klo_s = nullptr, *vt_s = nullptr;
  int* vlo_flag = nullptr;
  float* ksplit_ws = nullptr;
  void *vt32 = nullptr, *x4a_ws = nullptr;
  if (split) {
    vt_c = m->alloc_bytes((size_t)B * nt_c * 8192);
    khi_s = m->alloc_bytes((size_t)rows * 512); klo_s = m->alloc_bytes((size_t)rows * 512);
    vt_s = m->alloc_bytes((size_t)4 * B * nt_s * 8192);
    if (!vt_c || !khi_s || !klo_s || !vt_s) { ds2_set_error("memory_attention: workspace exhausted"); return DS2_ERR_STATE; }
    // V = raw memory (64-d), shared by the 4 layers.  Frame tokens of the bank are bf16 storage (lo plane == 0): the
    // split kernel verifies that on the fly (vlo_flag) and the attention kernel then skips the lo term for those tiles.
    ksplit_ws = reinterpret_cast<float*>(m->alloc_bytes(ksplit_bytes));
    if (!ksplit_ws) { ds2_set_error("memory_attention: workspace exhausted"); return DS2_ERR_STATE; }
    vlo_flag = reinterpret_cast<int*>(m->alloc_bytes(256));
    if (!vlo_flag) { ds2_set_error("memory_attention: workspace exhausted"); return DS2_ERR_STATE; }
    if (x4a) {
      vt32 = m->alloc_bytes((size_t)B * nt_c * 4096);
      x4a_ws = m->alloc_bytes(x4a_ws_bytes);
      if (!vt32 || !x4a_ws) { ds2_set_error("memory_attention: workspace exhausted"); return DS2_ERR_STATE; }
      if (bank_direct) {   // frame tokens here; the pointer tokens' slots together with their kin rows below (tiles zeroed first)
        TRY(bank_for_each(m, B, *bank, nullptr, nullptr, [&](const BankArgs& a) { return launch_bank_vt32(a, vt32, st); },
                          [&](const BankArgs&) { return (int)DS2_OK; }));
        const int t0 = (Nk - n_ptr_tok) / 32;
        if (nt_c > t0)
          DS2_CHECK_HIP(hipMemset2DAsync(reinterpret_cast<char*>(vt32) + (size_t)t0 * 4096, (size_t)nt_c * 4096, 0, (size_t)(nt_c - t0) * 4096, B, st));
      } else {
        TRY(launch_vt_pack32(memory, 64, B, Nk, vt32, st));
      }
    } else {
      TRY(launch_vt_split16(memory, 64, B, Nk, vt_c, 64, st, Nk - n_ptr_tok, vlo_flag, k_f16));   // (bf16x3k: fp16 planes)
    }
  }
  // output = curr + 0.1 * curr_pos (memory_attention.py:139-141); identical for every object in the tracking loop
  // (curr is the frame's feature, curr_pos the model constant); the general form takes per-object tokens / positions
  const float* cpos = curr_pos ? curr_pos : m->P("#vision_pos");
  if (shared0) {
    TRY(launch_add_bcast(curr, 256, cpos, 256, 0, 0.1f, x1, 256, TOK, 256, st));
  } else {
    for (int b = 0; b < B; ++b)
      TRY(launch_add_bcast(curr + (curr_shared ? 0 : (size_t)b * TOK * 256), 256,
                           cpos + ((curr_pos && !pos_shared) ? (size_t)b * TOK * 256 : 0), 256, 0, 0.1f, x + (size_t)b * TOK * 256, 256,
                           TOK, 256, st));
  }
  // k input of the cross attention: memory + memory_pos (pos_enc_at_cross_attn_keys, memory_attention.py:79)
  if (split) {   // emitted directly as the k_proj GEMM's operand planes
    ds2_model::ActPlanes kp;
    TRY(new_act_planes(m, kin, B * Nk, 64, &kp, st));
    if (bank_direct) {
      DS2_REQUIRE(kp.ld == 64, "memory_attention: key-input planes with row pitch %d", kp.ld);
      const unsigned char* slots = attention_x4a_vt_slot_table();
      DS2_REQUIRE(slots, "memory_attention: no V^T slot table");
      TRY(bank_for_each(m, B, *bank, nullptr, nullptr, [&](const BankArgs& a) { return launch_bank_kin(a, kp.hi, kp.lo, st); },
                        [&](const BankArgs& a) { return launch_bank_ptr_planes(a, m->P("#ptr_dim_t"), kp.hi, kp.lo, vt32, nt_c, slots, st); }));
    } else {
      TRY(launch_add_bcast_split(memory, 64, memory_pos, 64, 0, 1.0f, kp.hi, kp.lo, kp.ld, B * Nk, 64, st));
    }
  } else {
    TRY(launch_add_bcast(memory, 64, memory_pos, 64, 0, 1.0f, kin, 64, B * Nk, 64, st));
  }
  // norm1 of layers 1.. and the final norm are computed in the epilogue of the previous layer's fused MLP: their operand planes
  // live outside the per-layer workspace marks, two sets used alternately
  const bool fuse_ln = split;
  ds2_model::ActPlanes n1p[2] = {};
  if (fuse_ln)
    for (int i = 0; i < 2; ++i) {
      n1p[i].hi = reinterpret_cast<unsigned short*>(m->alloc_bytes((size_t)rows * 512));
      n1p[i].lo = reinterpret_cast<unsigned short*>(m->alloc_bytes((size_t)rows * 512));
      n1p[i].ld = 256;
      if (!n1p[i].hi || !n1p[i].lo) { ds2_set_error("memory_attention: workspace exhausted"); return DS2_ERR_STATE; }
    }
  bool n1_ready = false, final_done = false;
  const float sc = 1.0f / 16.0f;  // 1/sqrt(256)
  for (int l = 0; l < m->cfg.mem_attn_layers; ++l) {
    const std::string p = "memory_attention.layers." + std::to_string(l);
    const std::string ls = std::to_string(l);
    const size_t layer_mark = m->ws_top;   // operand planes emitted inside a layer die with it
    m->act_planes.erase(a);
    // -- self attention (RoPE on q,k).  Layer 0 sees the same input for all B objects: computed once.
    const bool once = (l == 0) && shared0;
    const int Bs = once ? 1 : B;
    const float* xin = once ? x1 : x;
    if (n1_ready) m->act_planes[t] = n1p[l & 1];   // emitted by the previous layer's MLP epilogue
    else TRY(layernorm(m, st, p + ".norm1", xin, t, Bs * TOK, 256, 1e-5f, DS2_ACT_NONE, true));
    // in_proj + the key rotation / fp16 plane + the V^T tiles as ONE kernel (gemm_qkvs.hip; where it does not apply: the GEMM,
    // k_rope_split and k_vt_split16).  Mode bf16x3k only: single fp16 planes, the values' lo plane is not read
    auto tpl = m->act_planes.find(t);
    const bool qkvf = split && k_f16 && !klo_planes && tpl != m->act_planes.end() && tpl->second.ld == 256 &&
                      m->P("@ma_qkv_w." + ls) && qkv_self_supported(Bs * TOK, 256, 256, TOK);
    const int ldq_s = qkvf ? 256 : 768;
    if (qkvf) {
      GemmPlanes wp;
      TRY(weight_planes(m->gctx, m->P("@ma_qkv_w." + ls), 768, 256, &wp, st));
      char ptag[96] = "";
      if (g_prof_gemm) snprintf(ptag, sizeof(ptag), "kern k_qkv_self %d %d %d", Bs * TOK, 768, 256);
      ProfScope _gp(ptag, st, g_prof_gemm);
      TRY(launch_qkv_self(tpl->second.hi, tpl->second.lo, tpl->second.ld, Bs * TOK, wp.hi, wp.lo, wp.ld, m->P("@ma_qkv_b." + ls), cis, TOK,
                          qkv, 256, khi_s, vt_s, st));
    } else {
      TRY(gemm(st, Bs * TOK, 768, 256, t, 256, m->P("@ma_qkv_w." + ls), 256, m->P("@ma_qkv_b." + ls), qkv, 768, DS2_ACT_NONE, nullptr, 0, 0, nullptr, true, m));
    }
    if (!split) TRY(launch_rope(qkv, 768, cis, Bs, TOK, TOK, TOK, st));   // (the bf16x3 kernel rotates q while loading it)
    if (split) {
      if (!qkvf) TRY(launch_rope_split(qkv + 256, 768, cis, Bs, TOK, TOK, TOK, khi_s, klo_planes ? klo_s : nullptr, st, k_f16));
      ProfScope _p("kernel.self_attention", st);
      ds2_model::ActPlanes sa_p{};
      float* xs = once ? x1 : x;               // the residual stream this self-attention updates
      if (!m->ma_fold_vo) TRY(new_act_planes(m, a, Bs * TOK, 256, &sa_p, st));   // consumer: out_proj GEMM
      if (!qkvf) TRY(launch_vt_split16(qkv + 512, 768, Bs, TOK, vt_s, 256, st, 0, nullptr, k_f16));   // all 256 value columns in one pass
      // (measured and not kept: norm2 of the written rows emitted as q_proj's operand planes in this kernel's epilogue - the epilogue's
      // 8-byte plane stores cost the kernel 16 us per launch, what the separate LayerNorm pass costs less its launch: +-0)
      if (m->ma_fold_vo)   // values already carry out_proj: the kernel adds its result to the residual stream in place
        TRY(launch_attention_w8(qkv, ldq_s, khi_s, klo_planes ? klo_s : nullptr, vt_s, xs, 256, Bs, TOK, TOK, sc, 256, st, nullptr,
                                nullptr, 0, 0, nullptr, cis, TOK, xs, 256, false, ksplit_ws, ksplit_bytes));
      else
        TRY(launch_attention_w8(qkv, ldq_s, khi_s, klo_planes ? klo_s : nullptr, vt_s, nullptr, 256, Bs, TOK, TOK, sc, 256, st, sa_p.hi,
                                sa_p.lo, sa_p.ld, 0, nullptr, cis, TOK, nullptr, 0, false, ksplit_ws, ksplit_bytes));
    } else {
      TRY(launch_rope(qkv + 256, 768, cis, Bs, TOK, TOK, TOK, st));
      AttnArgs sa{};
      sa.q = qkv; sa.k = qkv + 256; sa.v = qkv + 512; sa.o = a;
      sa.ldq = sa.ldk = sa.ldv = 768; sa.ldo = 256;
      sa.batch = Bs; sa.heads = 1; sa.D = 256; sa.DV = 256; sa.Lq = sa.Lk = TOK; sa.scale = sc;
      ProfScope _p("kernel.self_attention", st);
      TRY(launch_attention(sa, st));
      if (m->ma_fold_vo) {   // fp32 mode: x += attention (the values carry out_proj)
        float* xs = once ? x1 : x;
        TRY(launch_add_bcast(xs, 256, a, 256, 0, 1.0f, xs, 256, Bs * TOK, 256, st));
      }
    }
    // layer 0, shared input: the residual stream is still the same for every object up to the cross-attention's result, so
    // norm2 and q_proj run on ONE copy (TOK rows), the attention kernel reads those queries for every object, and the folded
    // value projection adds its result to x1 broadcast by row (r_mod) - no replication of x1, 15/16 of two passes saved
    const bool q_once = once && split && m->ma_fold_vo;
    if (once) {
      if (!m->ma_fold_vo) TRY(linear(m, st, p + ".self_attn.out_proj", TOK, 256, 256, a, 256, x1, 256, DS2_ACT_NONE, x1, 256));
      if (!q_once) TRY(launch_bcast_rows(x1, x, TOK * 256, B, st));   // one launch instead of B device copies
    } else if (!m->ma_fold_vo) {
      TRY(linear(m, st, p + ".self_attn.out_proj", rows, 256, 256, a, 256, x, 256, DS2_ACT_NONE, x, 256));
    }
    // -- cross attention to the memory bank.  V = v_proj(memory) is never materialised:
    //    softmax(QK^T) (M Wv^T + bv) = (softmax(QK^T) M) Wv^T + bv, so P.V runs in the 64-d memory space.
    m->act_planes.erase(a);   // self-attention planes are consumed; `a` is re-used below
    const int qrows = q_once ? TOK : rows;
    // norm2 -> q_proj -> RoPE -> scale -> fp16 Q fragments of the assembly attention as ONE kernel (gemm_qproj.hip; where it does
    // not apply: the LayerNorm pass, the GEMM and the attention's query pass)
    const float* qw = m->P(p + ".cross_attn_image.q_proj.weight");
    const bool qfuse = x4a && qw && qproj_x4a_supported(qrows, 256, 256) &&
                       m->P(p + ".norm2.weight") && m->P(p + ".norm2.bias");
    if (qfuse) {
      GemmPlanes qwp;
      TRY(weight_planes(m->gctx, qw, 256, 256, &qwp, st));
      char ptag[96] = "";
      if (g_prof_gemm) snprintf(ptag, sizeof(ptag), "kern k_qproj_x4a %d %d %d", qrows, 256, 256);
      ProfScope _gp(ptag, st, g_prof_gemm);
      TRY(launch_qproj_x4a(q_once ? x1 : x, 256, qrows, m->P(p + ".norm2.weight"), m->P(p + ".norm2.bias"), 1e-5f, qwp.hi, qwp.lo, qwp.ld,
                           m->P(p + ".cross_attn_image.q_proj.bias"), cis, TOK, sc, x4a_ws, q_once ? B : 1, st));
    } else {
      TRY(layernorm(m, st, p + ".norm2", q_once ? x1 : x, t, qrows, 256, 1e-5f, DS2_ACT_NONE, true));
      TRY(linear(m, st, p + ".cross_attn_image.q_proj", qrows, 256, 256, t, 256, q, 256));
    }
    if (!split) TRY(launch_rope(q, 256, cis, B, TOK, TOK, TOK, st));
    bool vofuse = false;
    const float *vo_po = nullptr, *vo_pml = nullptr;
    int vo_ns = 1;
    if (split) {
      // k_proj + RoPE + split fused: the GEMM epilogue rotates and emits the key planes, no fp32 K round trip
      TRY(gemm(st, B * Nk, 256, 64, kin, 64, m->P(p + ".cross_attn_image.k_proj.weight"), 64,
               m->P(p + ".cross_attn_image.k_proj.bias"), K, 256, DS2_ACT_NONE, nullptr, 0, 0, nullptr, true, m, true, cis, Nk,
               Nk - n_ptr_tok, TOK, !klo_planes, k_f16));   // (bf16x3k: ONE key plane, fp16)
      const ds2_model::ActPlanes kpl = m->act_planes[K];
      khi = kpl.hi; klo = klo_planes ? kpl.lo : nullptr;
      if (x4a && Nk % 32) {   // the last key tile of the last object reads past the plane: into the (unused) lo plane - finite values
        DS2_REQUIRE(reinterpret_cast<char*>(kpl.lo) == reinterpret_cast<char*>(kpl.hi) + (size_t)B * Nk * 512, "memory_attention: key planes not adjacent");
        DS2_CHECK_HIP(hipMemsetAsync(kpl.lo, 0, 32 * 512, st));
      }
      ProfScope _p("kernel.cross_attention", st);
      // the merge / normalisation of the attention's part(s) + the folded value / output projection + residual as ONE kernel (gemm_vo.hip;
      // where it does not apply: k_w8_merge -> planes -> K = 64 GEMM)
      vofuse = x4a && m->ma_fold_vo && m->P("#ma_cross_vo_w." + ls) && vo_merge_supported(rows, 64);
      if (vofuse) vo_ns = attention_x4a_parts(x4a_ws, B, TOK, Nk, &vo_po, &vo_pml);
      ds2_model::ActPlanes cp{};
      if (!vofuse) TRY(new_act_planes(m, a64, rows, 64, &cp, st));   // consumer: v_proj GEMM
      if (x4a)
        TRY(launch_attention_x4a(qfuse ? nullptr : q, 256, khi, vt32, B, TOK, Nk, sc, st, cp.hi, cp.lo, cp.ld, cis, TOK, q_once, x4a_ws, x4a_ws_bytes,
                                 !vofuse));
      else
        TRY(launch_attention_w8(q, 256, khi, klo, vt_c, nullptr, 64, B, TOK, Nk, sc, 64, st, cp.hi, cp.lo, cp.ld,
                                Nk - n_ptr_tok, vlo_flag, cis, TOK, nullptr, 0, q_once, ksplit_ws, ksplit_bytes));
    } else {
      TRY(linear(m, st, p + ".cross_attn_image.k_proj", B * Nk, 256, 64, kin, 64, K, 256));
      TRY(launch_rope(K, 256, cis, B, Nk, Nk - n_ptr_tok, TOK, st));
      AttnArgs ca{};
      ca.q = q; ca.k = K; ca.v = memory; ca.o = a64;
      ca.ldq = 256; ca.ldk = 256; ca.ldv = 64; ca.ldo = 64;
      ca.batch = B; ca.heads = 1; ca.D = 256; ca.DV = 64; ca.Lq = TOK; ca.Lk = Nk; ca.scale = sc;
      ProfScope _p("kernel.cross_attention", st);
      TRY(launch_attention(ca, st));
    }
    // out_proj(v_proj(P M)) folded on the host into ONE 64 -> 256 projection (constants.py fold_out_v: exact in real
    // arithmetic): x += (P M) (Wo Wv)^T + (Wo bv + bo) - one K = 64 GEMM instead of a K = 64 and a K = 256 one
    if (vofuse) {
      GemmPlanes vwp;
      TRY(weight_planes(m->gctx, m->P("#ma_cross_vo_w." + ls), 256, 64, &vwp, st));
      char ptag[96] = "";
      if (g_prof_gemm) snprintf(ptag, sizeof(ptag), "kern k_vo_merge %d %d %d", rows, 256, 64);
      ProfScope _gp(ptag, st, g_prof_gemm);
      TRY(launch_vo_merge(vo_po, vo_pml, vo_ns, rows, vwp.hi, vwp.lo, vwp.ld, m->P("#ma_cross_vo_b." + ls), q_once ? x1 : x, 256, q_once ? TOK : 0, x, 256, st));
    } else if (m->ma_fold_vo) {
      TRY(gemm(st, rows, 256, 64, a64, 64, m->P("#ma_cross_vo_w." + ls), 64, m->P("#ma_cross_vo_b." + ls), x, 256, DS2_ACT_NONE,
               q_once ? x1 : x, 256, q_once ? TOK : 0, nullptr, true, m));
    } else {
      TRY(linear(m, st, p + ".cross_attn_image.v_proj", rows, 256, 64, a64, 64, a, 256, DS2_ACT_NONE, nullptr, 0, 0, nullptr,
                 true));   // only consumer: out_proj GEMM
      TRY(linear(m, st, p + ".cross_attn_image.out_proj", rows, 256, 256, a, 256, x, 256, DS2_ACT_NONE, x, 256));
    }
    // -- FFN.  norm3 runs in the fused MLP's prologue when that kernel takes the layer in its two-fp16-term form (else as its
    //    own pass writing the operand planes; same bits either way - verified bit for bit in round 5)
    MlpArgs probe{};
    probe.rows = rows; probe.D = 256; probe.H = F; probe.ldx = 256; probe.ldw1 = 256; probe.ldw2 = F; probe.ldo = 256; probe.ldr = 256;
    const bool ln3_in = split && f16x2_enabled() && mlp256_supported(probe) &&
                        m->P(p + ".norm3.weight") && m->P(p + ".norm3.bias");
    const LnFuse lni{m->P(p + ".norm3.weight"), m->P(p + ".norm3.bias"), 1e-5f, nullptr, nullptr};
    if (!ln3_in) TRY(layernorm(m, st, p + ".norm3", x, t, rows, 256, 1e-5f, DS2_ACT_NONE, true));
    const bool last = l + 1 == m->cfg.mem_attn_layers;
    const std::string pn = last ? std::string("memory_attention.norm") : "memory_attention.layers." + std::to_string(l + 1) + ".norm1";
    const LnFuse lnf{m->P(pn + ".weight"), m->P(pn + ".bias"), 1e-5f, last ? out : nullptr, last ? nullptr : &n1p[(l + 1) & 1]};
    bool ln_done = false;
    TRY(mlp2(m, st, p + ".linear1", p + ".linear2", rows, F, ln3_in ? x : t, h, x, DS2_ACT_RELU, x, nullptr, false, f16x2_enabled(),
             (fuse_ln && lnf.w && lnf.b) ? &lnf : nullptr, &ln_done, ln3_in ? &lni : nullptr));
    n1_ready = ln_done && !last;
    final_done = ln_done && last;
    m->release(layer_mark);
  }
  if (!final_done) TRY(layernorm(m, st, "memory_attention.norm", x, out, rows, 256, 1e-5f));
  CHECK_PARAMS();
  return DS2_OK;
}

// ------------------------------------------------------------------------------------------------ A7 + A8
namespace {
// Attention module of the two-way transformer (transformer.py:239-284): projections + SDPA + out_proj.
// keys + key_pe (transformer.py:196-197,211) whose only consumers are projection GEMMs: in bf16x3 mode it is emitted
// directly as operand planes registered under `kpe` (one pass, re-used by every GEMM that reads it) instead of an fp32
// tensor that each GEMM would split again.
#ifndef DS2_MLP3_FUSED
#define DS2_MLP3_FUSED 1
#endif
#ifndef DS2_HEADS_TOK_STACK
#define DS2_HEADS_TOK_STACK 1
#endif
#ifndef DS2_HEADS_LN_KPE
#define DS2_HEADS_LN_KPE 1
#endif
#ifndef DS2_HEADS_O_PLANES
#define DS2_HEADS_O_PLANES 1
#endif
#ifndef DS2_HEADS_LN_PE
#define DS2_HEADS_LN_PE 1
#endif
#ifndef DS2_KPE_PLANES
#define DS2_KPE_PLANES 1
#endif
int keys_plus_pe(ds2_model* m, hipStream_t st, const float* keys, const float* dense_pe, float* kpe, int rows) {
  if (DS2_KPE_PLANES && ds2_split_mode()) {
    ds2_model::ActPlanes kp;
    TRY(new_act_planes(m, kpe, rows, 256, &kp, st));
    return launch_add_bcast_split(keys, 256, dense_pe, 256, TOK, 1.0f, kp.hi, kp.lo, kp.ld, rows, 256, st);
  }
  m->act_planes.erase(kpe);
  return launch_add_bcast(keys, 256, dense_pe, 256, TOK, 1.f, kpe, 256, rows, 256, st);
}
// q_in [B*Lq,256], k_in/v_in [B*Lk,256]; result (+ optional residual R) -> out [B*Lq,256].
// pre (optional): projections the caller has already computed (stacked GEMMs over a shared input) - pointer + row stride per
// operand; a null pointer = project here from q_in / k_in / v_in.
struct SamPre {
  const float* q = nullptr; int ldq = 0;
  const float* k = nullptr; int ldk = 0;
  const float* v = nullptr; int ldv = 0;
};
int sam_attention(ds2_model* m, hipStream_t st, const std::string& p, int B, int Lq, int Lk, int internal,
                  const float* q_in, const float* k_in, const float* v_in, float* out, const float* R, const SamPre& pre = SamPre()) {
  const size_t mark = m->ws_top;
  const float *q = pre.q, *k = pre.k, *v = pre.v;
  if (!q) {
    ALLOC(qb, (size_t)B * Lq * internal);
    TRY(linear(m, st, p + ".q_proj", B * Lq, internal, 256, q_in, 256, qb, internal));
    q = qb;
  }
  if (!k) {
    ALLOC(kb, (size_t)B * Lk * internal);
    TRY(linear(m, st, p + ".k_proj", B * Lk, internal, 256, k_in, 256, kb, internal));
    k = kb;
  }
  if (!v) {
    ALLOC(vb, (size_t)B * Lk * internal);
    TRY(linear(m, st, p + ".v_proj", B * Lk, internal, 256, v_in, 256, vb, internal));
    v = vb;
  }
  ALLOC(o, (size_t)B * Lq * internal);
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = o;
  a.ldq = pre.q ? pre.ldq : internal;
  a.ldk = pre.k ? pre.ldk : internal;
  a.ldv = pre.v ? pre.ldv : internal;
  a.ldo = internal;
  a.batch = B; a.heads = 8; a.D = a.DV = internal / 8; a.Lq = Lq; a.Lk = Lk;
  a.scale = 1.0f / sqrtf((float)a.D);
  // image-side queries (image -> token): the tile kernel writes its result as the operand planes of out_proj (no fp32 `o`,
  // no split pre-pass); the few-query kernels of the token side have no plane output
  if (DS2_HEADS_O_PLANES && ds2_split_mode() && Lq >= 1024 && a.D == 16) {
    ds2_model::ActPlanes op;
    TRY(new_act_planes(m, o, B * Lq, internal, &op, st));
    a.o_hi = op.hi; a.o_lo = op.lo; a.ldop = op.ld;
  }
  TRY(launch_attention(a, st));
  TRY(linear(m, st, p + ".out_proj", B * Lq, 256, internal, o, internal, out, 256, DS2_ACT_NONE, R, 256));
  m->release(mark);
  return DS2_OK;
}
int mlp3(ds2_model* m, hipStream_t st, const std::string& p, int M, const float* A, int lda, int hidden, int n_out, float* out,
         int ldc, int last_act) {
  if (DS2_MLP3_FUSED && hidden == 256 && M <= 64 && n_out <= 256)   // one launch, exact fp32 (kernels.hip k_mlp3_256)
    return launch_mlp3_256(A, lda, m->P(p + ".layers.0.weight"), m->P(p + ".layers.0.bias"), m->P(p + ".layers.1.weight"),
                           m->P(p + ".layers.1.bias"), m->P(p + ".layers.2.weight"), m->P(p + ".layers.2.bias"), n_out, out, ldc,
                           last_act, M, st);
  const size_t mark = m->ws_top;
  ALLOC(h1, (size_t)M * hidden);
  ALLOC(h2, (size_t)M * hidden);
  TRY(linear(m, st, p + ".layers.0", M, hidden, 256, A, lda, h1, hidden, DS2_ACT_RELU));
  TRY(linear(m, st, p + ".layers.1", M, hidden, hidden, h1, hidden, h2, hidden, DS2_ACT_RELU));
  TRY(linear(m, st, p + ".layers.2", M, n_out, hidden, h2, hidden, out, ldc, last_act));
  m->release(mark);
  return DS2_OK;
}
}  // namespace

extern "C" int ds2_sam_heads(ds2_model* m, int32_t B, const float* pix_feat, int32_t pix_bcast, int32_t add_no_mem_embed,
                             const float* fpn0, const float* fpn1, const float* point_coords, const int32_t* point_labels,
                             int32_t P, int32_t multimask, float* low_res, float* obj_ptr, float* obj_logits, float* ious,
                             void* stream) {
  return ds2_sam_heads_mask(m, B, pix_feat, pix_bcast, add_no_mem_embed, fpn0, fpn1, point_coords, point_labels, P, nullptr,
                            multimask, low_res, obj_ptr, obj_logits, ious, stream);
}

// MaskDecoder.forward as a module of its own (ds2_mask_decoder): the decoder core below runs on the CALLER's sparse / dense
// prompt embeddings and image_pe and returns all four masks / IoUs / mask tokens before the _forward_sam_heads glue.
struct SamDecoderIO {
  const float* sparse_in;   // [B, Ns, 256]
  int Ns;
  const float* dense_in;    // [B*4096, 256] token-major
  const float* image_pe;    // [4096, 256]
  float *masks4, *iou4, *mask_tokens;   // [B,4,65536], [B,4], [B,4,256]
};
static int sam_heads_impl(ds2_model* m, int32_t B, const float* pix_feat, int32_t pix_bcast, int32_t add_no_mem_embed,
                          const float* fpn0, const float* fpn1, const float* point_coords, const int32_t* point_labels,
                          int32_t P, const float* mask_inputs, int32_t multimask, float* low_res, float* obj_ptr,
                          float* obj_logits, float* ious, void* stream, const SamDecoderIO* io);

extern "C" int ds2_sam_heads_mask(ds2_model* m, int32_t B, const float* pix_feat, int32_t pix_bcast, int32_t add_no_mem_embed,
                                  const float* fpn0, const float* fpn1, const float* point_coords,
                                  const int32_t* point_labels, int32_t P, const float* mask_inputs, int32_t multimask,
                                  float* low_res, float* obj_ptr, float* obj_logits, float* ious, void* stream) {
  DS2_REQUIRE(m && m->finalized && B > 0 && pix_feat && fpn0 && fpn1 && low_res && obj_ptr && obj_logits,
              "ds2_sam_heads: bad argument");
  return sam_heads_impl(m, B, pix_feat, pix_bcast, add_no_mem_embed, fpn0, fpn1, point_coords, point_labels, P, mask_inputs,
                        multimask, low_res, obj_ptr, obj_logits, ious, stream, nullptr);
}

extern "C" int ds2_mask_decoder(ds2_model* m, int32_t B, const float* image_embeddings, const float* image_pe,
                                const float* sparse, int32_t Ns, const float* dense, const float* feat_s0, const float* feat_s1,
                                float* masks4, float* iou4, float* mask_tokens, float* obj_logits, void* stream) {
  DS2_REQUIRE(m && m->finalized && B > 0 && image_embeddings && image_pe && dense && feat_s0 && feat_s1 && masks4 && iou4 &&
              mask_tokens && obj_logits && Ns >= 0 && Ns <= 257 && (Ns == 0 || sparse), "ds2_mask_decoder: bad argument");
  SamDecoderIO io{sparse, Ns, dense, image_pe, masks4, iou4, mask_tokens};
  return sam_heads_impl(m, B, image_embeddings, 0, 0, feat_s0, feat_s1, nullptr, nullptr, 0, nullptr, 1, nullptr, nullptr,
                        obj_logits, nullptr, stream, &io);
}

extern "C" int ds2_prompt_encoder(ds2_model* m, int32_t B, const float* point_coords, const int32_t* point_labels, int32_t P,
                                  int32_t pad, const float* mask_inputs, float* sparse, float* dense, void* stream) {
  DS2_REQUIRE(m && m->finalized && B > 0 && P >= 0 && P <= 256 && (pad == 0 || pad == 1) && (P == 0 || (point_coords && point_labels)),
              "ds2_prompt_encoder: bad argument");
  const int Ns = P + (P > 0 ? pad : 0);      // no points: no padding point either (prompt_encoder.py:155-160) - an empty [B,0,256]
  DS2_REQUIRE(Ns == 0 || sparse, "ds2_prompt_encoder: sparse output missing");   // sparse tensor: mask-only / empty prompts
  ModelScope _dg(m);
  hipStream_t st = (hipStream_t)stream;
  TRY(m->require(((size_t)B * (6 + Ns) * 256 + (size_t)TOK * 256) * 4 + (1u << 20), st));
  if (Ns > 0) {
    ALLOC(tokens, (size_t)B * (6 + Ns) * 256);
    TRY(launch_prompt_tokens(m->P("@out_tokens6"), m->P("sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"),
                             m->P("@point_emb4"), m->P("sam_prompt_encoder.not_a_point_embed.weight"), point_coords, point_labels,
                             B, P, 1024.f, tokens, st, pad));
    DS2_CHECK_HIP(hipMemcpy2DAsync(sparse, (size_t)Ns * 1024, tokens + 6 * 256, (size_t)(6 + Ns) * 1024, (size_t)Ns * 1024, B,
                                   hipMemcpyDeviceToDevice, st));
  }
  if (dense) {
    const size_t rows = (size_t)B * TOK;
    if (mask_inputs) {   // mask_downscaling (prompt_encoder.py:53-61,97-100)
      const std::string pe = "sam_prompt_encoder.mask_downscaling.";
      const float* prm[10] = {m->P(pe + "0.weight"), m->P(pe + "0.bias"), m->P(pe + "1.weight"), m->P(pe + "1.bias"),
                              m->P(pe + "3.weight"), m->P(pe + "3.bias"), m->P(pe + "4.weight"), m->P(pe + "4.bias"),
                              m->P(pe + "6.weight"), m->P(pe + "6.bias")};
      for (int i = 0; i < 10; ++i) DS2_REQUIRE(prm[i], "ds2_prompt_encoder: mask_downscaling parameter %d missing", i);
      ALLOC(zero, (size_t)TOK * 256);
      DS2_CHECK_HIP(hipMemsetAsync(zero, 0, (size_t)TOK * 256 * 4, st));
      TRY(launch_mask_downscale_add(mask_inputs, prm, zero, 1, dense, B, st));
    } else {             // no_mask_embed broadcast over the grid (:167-169)
      DS2_CHECK_HIP(hipMemsetAsync(dense, 0, rows * 256 * 4, st));
      TRY(launch_add_rowvec(dense, 256, m->P("@dense_vec"), dense, 256, (int)rows, 256, st));
    }
  }
  CHECK_PARAMS();
  return DS2_OK;
}

static int sam_heads_impl(ds2_model* m, int32_t B, const float* pix_feat, int32_t pix_bcast, int32_t add_no_mem_embed,
                          const float* fpn0, const float* fpn1, const float* point_coords, const int32_t* point_labels,
                          int32_t P, const float* mask_inputs, int32_t multimask, float* low_res, float* obj_ptr,
                          float* obj_logits, float* ious, void* stream, const SamDecoderIO* io) {
  ModelScope _dg(m);
  DS2_REQUIRE(P >= 0 && P <= 256 && (P == 0 || (point_coords && point_labels)), "ds2_sam_heads: bad prompt");
  if (io) P = io->Ns;
  hipStream_t st = (hipStream_t)stream;
  ProfScope _ps("stage.sam_heads", st);
  const int rows = B * TOK;
  const size_t Tmax = 6 + (size_t)(P > 0 ? P : 1) + 1;      // decoder tokens: 6 output tokens + prompt points + the padding point
  const size_t need = ((size_t)rows * 256 * 9 + (size_t)B * 16384 * (64 + 128) + (size_t)B * 4 * 65536 + (size_t)B * Tmax * (2048 + 256 * 10 + 1024) * 2) * 4 +
                      (size_t)6 * rows * 256 * 4 /* keys / keys + pe operand planes, one set per use */ + (8u << 20);
  TRY(m->require(need, st));
  const std::string md = "sam_mask_decoder", tr = md + ".transformer";
  // no prompt => the reference feeds one dummy point labelled -1 (sam2_base.py:298-301)
  int Pe = P;
  const float* coords = point_coords;
  const int* labels = point_labels;
  const int pad = io ? 0 : 1;
  if (P == 0 && !io) {
    Pe = 1;
    float* zc = m->alloc((size_t)B * 2);
    int* ml = reinterpret_cast<int*>(m->alloc_bytes((size_t)B * 4));
    if (!zc || !ml) { ds2_set_error("ds2_sam_heads: workspace exhausted"); return DS2_ERR_STATE; }
    DS2_CHECK_HIP(hipMemsetAsync(zc, 0, (size_t)B * 8, st));
    DS2_CHECK_HIP(hipMemsetAsync(ml, 0xFF, (size_t)B * 4, st));
    coords = zc;
    labels = ml;
  }
  const int T = 6 + Pe + pad;
  ALLOC(tokens, (size_t)B * T * 256);
  TRY(launch_prompt_tokens(m->P("@out_tokens6"), m->P("sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"),
                           m->P("@point_emb4"), m->P("sam_prompt_encoder.not_a_point_embed.weight"), coords, labels, B, Pe,
                           1024.f, tokens, st, pad, io ? io->sparse_in : nullptr));
  // src = image_embeddings + dense_prompt (no_mask_embed broadcast)  (mask_decoder.py:203)
  ALLOC(keys, (size_t)rows * 256);
  ALLOC(kpe, (size_t)rows * 256);      // keys + key_pe
  const float* const kpe_key = kpe;
  bool kpe_ready = false;              // the planes of keys + key_pe are already written (keys init / the previous norm4)
  const float* src = pix_feat;
  if (pix_bcast && add_no_mem_embed) {   // directly_add_no_mem_embed (sam2_base.py:651-657)
    ALLOC(pm, (size_t)TOK * 256);
    TRY(launch_add_rowvec(pix_feat, 256, m->P("no_mem_embed"), pm, 256, TOK, 256, st));
    src = pm;
  }
  DS2_REQUIRE(pix_bcast || !add_no_mem_embed, "ds2_sam_heads: add_no_mem_embed requires pix_bcast");
  if (io) {            // src = image_embeddings + the caller's dense prompt embeddings (mask_decoder.py:203)
    TRY(launch_add_bcast(pix_feat, 256, io->dense_in, 256, 0, 1.f, keys, 256, rows, 256, st));
  } else if (mask_inputs) {   // dense prompt = mask_downscaling(mask) instead of no_mask_embed (prompt_encoder.py:97-100,163-168)
    const std::string pe = "sam_prompt_encoder.mask_downscaling.";
    const float* prm[10] = {m->P(pe + "0.weight"), m->P(pe + "0.bias"), m->P(pe + "1.weight"), m->P(pe + "1.bias"),
                            m->P(pe + "3.weight"), m->P(pe + "3.bias"), m->P(pe + "4.weight"), m->P(pe + "4.bias"),
                            m->P(pe + "6.weight"), m->P(pe + "6.bias")};
    for (int i = 0; i < 10; ++i) DS2_REQUIRE(prm[i], "ds2_sam_heads: mask_downscaling parameter %d missing", i);
    TRY(launch_mask_downscale_add(mask_inputs, prm, src, pix_bcast ? 1 : 0, keys, B, st));
  } else if (DS2_HEADS_LN_KPE && DS2_KPE_PLANES && ds2_split_mode()) {
    // keys (fp32 + operand planes: v_proj / the upscaling GEMM read them without a split pre-pass) and the planes of keys + key_pe
    ds2_model::ActPlanes kp, pp;
    TRY(new_act_planes(m, keys, rows, 256, &kp, st));
    TRY(new_act_planes(m, kpe_key, rows, 256, &pp, st));
    TRY(launch_sam_keys_init(src, pix_bcast ? TOK : 0, m->P("@dense_vec"), m->P("#dense_pe"), TOK, keys, kp.hi, kp.lo, pp.hi, pp.lo, rows, st));
    kpe_ready = true;
  } else if (pix_bcast) {
    for (int b = 0; b < B; ++b)
      TRY(launch_add_rowvec(src, 256, m->P("@dense_vec"), keys + (size_t)b * TOK * 256, 256, TOK, 256, st));
  } else {
    TRY(launch_add_rowvec(pix_feat, 256, m->P("@dense_vec"), keys, 256, rows, 256, st));
  }
  ALLOC(queries, (size_t)B * T * 256);
  ALLOC(qpe, (size_t)B * T * 256);     // queries + query_pe
  ALLOC(tmpq, (size_t)B * T * 256);
  ALLOC(tmpk, (size_t)rows * 256);
  ALLOC(kq, (size_t)rows * 256);       // [k_proj of token->image | q_proj of image->token] of the current block
  ALLOC(hid, (size_t)B * T * 2048);
  const float* dense_pe = io ? io->image_pe : m->P("#dense_pe");
  const int BT = B * T;
  // queries = LayerNorm(tmpq); qpe = queries + query_pe (transformer.py:199-201,207-208) - one launch
  auto norm_pe = [&](const std::string& name) -> int {
    if (!DS2_HEADS_LN_PE) {
      TRY(layernorm(m, st, name, tmpq, queries, BT, 256, 1e-5f));
      return launch_add_bcast(queries, 256, tokens, 256, 0, 1.f, qpe, 256, BT, 256, st);
    }
    const float* w = m->P(name + ".weight");
    const float* b = m->P(name + ".bias");
    if (!w || !b) { ds2_set_error("missing parameter '%s'", name.c_str()); return DS2_ERR_STATE; }
    return launch_layernorm_add(tmpq, 256, w, b, queries, 256, tokens, qpe, BT, 256, 1e-5f, st);
  };
  // token-side projections stacked per shared input (DS2_HEADS_TOK_STACK): tq = [i2t.k | next self.q | next self.k] resp.
  // [i2t.k | final.q] of queries + query_pe, tv = [i2t.v | next self.v] of queries, s0 = layer 0's q | k | v of the prompt tokens
  ALLOC(tq, (size_t)BT * 640);
  ALLOC(tv, (size_t)BT * 384);
  auto stacked = [&](const std::string& name, int N, const float* A, float* out, bool* done) -> int {   // *done = false: not available
    *done = DS2_HEADS_TOK_STACK && m->Pbytes(name + "_w") == (size_t)N * 256 * 4 && m->Pbytes(name + "_b") == (size_t)N * 4;
    if (!*done) return DS2_OK;   // (Pbytes first: P() of an absent name would flag a missing parameter)
    const float* w = m->P(name + "_w");
    return gemm(st, BT, N, 256, A, 256, w, 256, m->P(name + "_b"), out, N, DS2_ACT_NONE, nullptr, 0, 0, nullptr, true, m);
  };
  bool have_tq = false, have_tv = false;   // tq / tv hold the projections of the current queries
  // TwoWayTransformer (transformer.py:91-131) with TwoWayAttentionBlock (:182-215)
  for (int l = 0; l < 2; ++l) {
    const std::string p = tr + ".layers." + std::to_string(l);
    if (l == 0) {   // skip_first_layer_pe: queries = self_attn(q=k=v=queries), no residual
      SamPre sp;
      bool s0 = false;
      TRY(stacked("@sam_s0_qkv", 768, tokens, hid, &s0));   // (hid: free until the MLP)
      if (s0) sp = SamPre{hid, 768, hid + 256, 768, hid + 512, 768};
      TRY(sam_attention(m, st, p + ".self_attn", B, T, T, 256, tokens, tokens, tokens, tmpq, nullptr, sp));
    } else {
      if (!DS2_HEADS_LN_PE) TRY(launch_add_bcast(queries, 256, tokens, 256, 0, 1.f, qpe, 256, BT, 256, st));
      // (DS2_HEADS_LN_PE: qpe = queries + query_pe was written with norm3 of the previous layer; queries are unchanged since)
      SamPre sp;
      if (have_tq) { sp.q = tq + 128; sp.ldq = 640; sp.k = tq + 384; sp.ldk = 640; }
      if (have_tv) { sp.v = tv + 128; sp.ldv = 384; }
      TRY(sam_attention(m, st, p + ".self_attn", B, T, T, 256, qpe, qpe, queries, tmpq, queries, sp));
    }
    // tokens -> image
    TRY(norm_pe(p + ".norm1"));
    if (!kpe_ready) TRY(keys_plus_pe(m, st, keys, dense_pe, kpe, rows));
    const float* kq_w = (m->Pbytes("@sam_kq_w." + std::to_string(l))) ? m->P("@sam_kq_w." + std::to_string(l)) : nullptr;
    if (kq_w)   // k_proj of this attention and q_proj of the image->token attention below read the same keys + key_pe: one GEMM
      TRY(gemm(st, rows, 256, 256, kpe, 256, kq_w, 256, m->P("@sam_kq_b." + std::to_string(l)), kq, 256, DS2_ACT_NONE, nullptr, 0, 0,
               nullptr, true, m));
    {
      SamPre sp;
      if (kq_w) { sp.k = kq; sp.ldk = 256; }
      TRY(sam_attention(m, st, p + ".cross_attn_token_to_image", B, T, TOK, 128, qpe, kpe, keys, tmpq, queries, sp));
    }
    TRY(layernorm(m, st, p + ".norm2", tmpq, queries, BT, 256, 1e-5f));
    // MLP
    TRY(linear(m, st, p + ".mlp.layers.0", BT, 2048, 256, queries, 256, hid, 2048, DS2_ACT_RELU));
    TRY(linear(m, st, p + ".mlp.layers.1", BT, 256, 2048, hid, 2048, tmpq, 256, DS2_ACT_NONE, queries, 256));
    // image -> tokens
    TRY(norm_pe(p + ".norm3"));
    TRY(stacked("@sam_tq." + std::to_string(l), l == 0 ? 640 : 256, qpe, tq, &have_tq));
    have_tv = false;
    if (l == 0) TRY(stacked("@sam_tv.0", 384, queries, tv, &have_tv));
    {
      SamPre sp;
      if (kq_w) { sp.q = kq + 128; sp.ldq = 256; }
      if (have_tq) { sp.k = tq; sp.ldk = l == 0 ? 640 : 256; }
      if (have_tv) { sp.v = tv; sp.ldv = 384; }
      TRY(sam_attention(m, st, p + ".cross_attn_image_to_token", B, TOK, T, 128, kpe, qpe, queries, tmpk, keys, sp));
    }
    // keys = norm4(...) and, in the same launch, the operand planes of keys + key_pe for the next block / the final attention
    kpe_ready = false;
    if (DS2_HEADS_LN_KPE && DS2_KPE_PLANES && ds2_split_mode()) {
      const float* w4 = m->P(p + ".norm4.weight");
      const float* b4 = m->P(p + ".norm4.bias");
      if (!w4 || !b4) { ds2_set_error("missing parameter '%s.norm4'", p.c_str()); return DS2_ERR_STATE; }
      ds2_model::ActPlanes kp, k0;
      TRY(new_act_planes(m, kpe, rows, 256, &kp, st));
      TRY(new_act_planes(m, keys, rows, 256, &k0, st));   // (replaces the registration of the previous keys)
      TRY(launch_layernorm_add_split(tmpk, 256, w4, b4, keys, 256, dense_pe, TOK, kp.hi, kp.lo, kp.ld, rows, 256, 1e-5f, st, k0.hi, k0.lo));
      kpe_ready = true;
    } else {
      TRY(layernorm(m, st, p + ".norm4", tmpk, keys, rows, 256, 1e-5f));
    }
  }
  if (!DS2_HEADS_LN_PE) TRY(launch_add_bcast(queries, 256, tokens, 256, 0, 1.f, qpe, 256, BT, 256, st));
  if (!kpe_ready) TRY(keys_plus_pe(m, st, keys, dense_pe, kpe, rows));
  {
    SamPre sp;
    if (have_tq) { sp.q = tq + 128; sp.ldq = 256; }   // (queries + query_pe unchanged since norm3 of the last block)
    TRY(sam_attention(m, st, tr + ".final_attn_token_to_image", B, T, TOK, 128, qpe, kpe, keys, tmpq, queries, sp));
  }
  float* hs = queries;
  TRY(layernorm(m, st, tr + ".norm_final_attn", tmpq, hs, BT, 256, 1e-5f));
  // upscaling + hypernetworks (mask_decoder.py:216-235)
  float* g1 = tmpk;  // [rows,256]
  TRY(gemm(st, rows, 256, 256, keys, 256, m->P("@up1_w"), 256, m->P("@up1_b"), g1, 256, DS2_ACT_NONE, nullptr, 0, 0, nullptr, true, m));
  ALLOC(u1, (size_t)B * 16384 * 64);
  {   // u1 feeds the second upscaling GEMM only: written as its operand planes in the split modes
    ds2_model::ActPlanes up{};
    if (DS2_HEADS_O_PLANES && ds2_split_mode()) TRY(new_act_planes(m, u1, B * 16384, 64, &up, st));
    TRY(launch_upscale1(g1, fpn1, m->P(md + ".output_upscaling.1.weight"), m->P(md + ".output_upscaling.1.bias"), u1, B, st, up.hi, up.lo));
  }
  ALLOC(g2, (size_t)B * 16384 * 128);
  TRY(gemm(st, B * 16384, 128, 64, u1, 64, m->P("@up2_w"), 64, m->P("@up2_b"), g2, 128, DS2_ACT_NONE, nullptr, 0, 0, nullptr, true, m));
  ALLOC(hyper, (size_t)B * 128);
  ALLOC(iou4, (size_t)B * 4);
  const bool heads_batched = DS2_MLP3_FUSED && B <= 64;
  if (heads_batched) {   // the six MLPs that read the output tokens (4 hypernetworks, IoU, object score) as ONE launch
    Mlp3Batch jb{};
    auto job = [&](int i, const std::string& p, const float* A, int n_out, float* out, int ldc, int last_act) {
      jb.job[i] = Mlp3Job{A, m->P(p + ".layers.0.weight"), m->P(p + ".layers.0.bias"), m->P(p + ".layers.1.weight"),
                          m->P(p + ".layers.1.bias"), m->P(p + ".layers.2.weight"), m->P(p + ".layers.2.bias"), out,
                          T * 256, n_out, ldc, last_act};
    };
    for (int i = 0; i < 4; ++i)
      job(i, md + ".output_hypernetworks_mlps." + std::to_string(i), hs + (2 + i) * 256, 32, hyper + i * 32, 128, DS2_ACT_NONE);
    job(4, md + ".iou_prediction_head", hs + 256, 4, iou4, 4, DS2_ACT_SIGMOID);
    job(5, md + ".pred_obj_score_head", hs, 1, obj_logits, 1, DS2_ACT_NONE);
    TRY(launch_mlp3_256_batch(jb, 6, B, st));
  } else {
    for (int i = 0; i < 4; ++i)
      TRY(mlp3(m, st, md + ".output_hypernetworks_mlps." + std::to_string(i), B, hs + (2 + i) * 256, T * 256, 256, 32,
               hyper + i * 32, 128, DS2_ACT_NONE));
  }
  ALLOC(masks4, (size_t)B * 4 * 65536);
  TRY(launch_upscale2_masks(g2, fpn0, hyper, masks4, B, st));
  if (!heads_batched) {
    TRY(mlp3(m, st, md + ".iou_prediction_head", B, hs + 256, T * 256, 256, 4, iou4, 4, DS2_ACT_SIGMOID));
    TRY(mlp3(m, st, md + ".pred_obj_score_head", B, hs, T * 256, 256, 1, obj_logits, 1, DS2_ACT_NONE));
  }
  if (io) {   // MaskDecoder.predict_masks' results (mask_decoder.py:163-259); forward()'s slicing is the caller's (Python module)
    DS2_CHECK_HIP(hipMemcpyAsync(io->masks4, masks4, (size_t)B * 4 * 65536 * 4, hipMemcpyDeviceToDevice, st));
    DS2_CHECK_HIP(hipMemcpyAsync(io->iou4, iou4, (size_t)B * 4 * 4, hipMemcpyDeviceToDevice, st));
    DS2_CHECK_HIP(hipMemcpy2DAsync(io->mask_tokens, 4 * 1024, hs + 2 * 256, (size_t)T * 1024, 4 * 1024, B, hipMemcpyDeviceToDevice, st));
    CHECK_PARAMS();
    return DS2_OK;
  }
  ALLOC(sel_tok, (size_t)B * 256);
  TRY(launch_select_masks(masks4, iou4, obj_logits, hs, T * 256, multimask, m->cfg.dynamic_multimask_stability_delta,
                          m->cfg.dynamic_multimask_stability_thresh, low_res, sel_tok, ious, B, st));
  // object pointer (sam2_base.py:373-387)
  TRY(mlp3(m, st, "obj_ptr_proj", B, sel_tok, 256, 256, 256, obj_ptr, 256, DS2_ACT_NONE));
  TRY(launch_ptr_gate(obj_ptr, obj_logits, m->P("no_obj_ptr"), B, 256, st));
  CHECK_PARAMS();
  return DS2_OK;
}

// ------------------------------------------------------------------------------------------------ A13
// mask source: low-res logits (upsampled + sigmoid|binarize, *scale+bias: the tracking loop) or full-resolution masks
// (optionally through a plain sigmoid: MemoryEncoder.forward's own contract); pix_feat shared by the objects or one per
// object; output bf16 with no_obj_embed_spatial (tracking loop) or the module's raw fp32 features.
static int memory_encoder_impl(ds2_model* m, int32_t B, const float* fpn2, bool pix_shared, const float* low_res,
                               const float* masks_hi, int hi_sigmoid, const float* obj_logits, int32_t binarize,
                               uint16_t* maskmem_bf16, float* out_f32, void* stream);
extern "C" int ds2_memory_encoder(ds2_model* m, int32_t B, const float* fpn2, const float* low_res, const float* obj_logits,
                                  int32_t binarize, uint16_t* maskmem_bf16, void* stream) {
  DS2_REQUIRE(low_res && obj_logits && maskmem_bf16, "ds2_memory_encoder: bad argument");
  return memory_encoder_impl(m, B, fpn2, true, low_res, nullptr, 0, obj_logits, binarize, maskmem_bf16, nullptr, stream);
}
extern "C" int ds2_memory_encoder_ex(ds2_model* m, int32_t B, const float* pix_feat, int32_t pix_shared, const float* masks,
                                     int32_t skip_mask_sigmoid, float* vision_features, void* stream) {
  DS2_REQUIRE(masks && vision_features, "ds2_memory_encoder_ex: bad argument");
  return memory_encoder_impl(m, B, pix_feat, pix_shared != 0, nullptr, masks, skip_mask_sigmoid ? 0 : 1, nullptr, 0, nullptr,
                             vision_features, stream);
}
static int memory_encoder_impl(ds2_model* m, int32_t B, const float* fpn2, bool pix_shared, const float* low_res,
                               const float* masks_hi, int hi_sigmoid, const float* obj_logits, int32_t binarize,
                               uint16_t* maskmem_bf16, float* out_f32, void* stream) {
  DS2_REQUIRE(m && m->finalized && B > 0 && fpn2, "ds2_memory_encoder: bad argument");
  ModelScope _dg(m);
  hipStream_t st = (hipStream_t)stream;
  ProfScope _ps("stage.memory_encoder", st);
  const int rows = B * TOK;
  const size_t need = ((size_t)B * 1048576 * 3 + (size_t)B * 16384 * (144 + 64 * 2) + (size_t)rows * (576 + 256 * 5 + 1024 + 64) + (size_t)TOK * 256) * 4 + (size_t)rows * (256 * 4 + 1024 * 2) * 4 + 2 * mlp256_part_bytes(rows, 1024) + (8u << 20);
  TRY(m->require(need, st));
  const std::string me = "memory_encoder", ds = me + ".mask_downsampler.encoder.";
  ALLOC(c1, (size_t)B * 262144 * 4);
  // low-res logits: upsample + sigmoid / binarise + first conv stage in one kernel - the 1024^2 mask is never materialised
  if (!masks_hi) {
    TRY(launch_mask_up_conv1(low_res, 256, 1024, binarize ? 1 : 0, m->cfg.sigmoid_scale_for_mem_enc, m->cfg.sigmoid_bias_for_mem_enc,
                             m->P(ds + "0.weight"), m->P(ds + "0.bias"), m->P(ds + "1.weight"), m->P(ds + "1.bias"), c1, B, st));
  } else {
    ALLOC(high, (size_t)B * 1048576);
    if (masks_hi && !hi_sigmoid) {
      DS2_CHECK_HIP(hipMemcpyAsync(high, masks_hi, (size_t)B * 1048576 * 4, hipMemcpyDeviceToDevice, st));
    } else if (masks_hi) {   // identity resample + sigmoid (memory_encoder.py:166-167)
      TRY(launch_mask_upsample_transform(masks_hi, high, B, 1024, 1024, 0, 1.f, 0.f, st));
    } else {
      TRY(launch_mask_upsample_transform(low_res, high, B, 256, 1024, binarize ? 1 : 0, m->cfg.sigmoid_scale_for_mem_enc,
                                         m->cfg.sigmoid_bias_for_mem_enc, st));
    }
    TRY(launch_conv3x3s2_small(high, m->P(ds + "0.weight"), m->P(ds + "0.bias"), m->P(ds + "1.weight"), m->P(ds + "1.bias"), c1,
                               B, 1024, 1, 4, st));
  }
  ALLOC(c2, (size_t)B * 65536 * 16);
  TRY(launch_conv3x3s2_small(c1, m->P(ds + "3.weight"), m->P(ds + "3.bias"), m->P(ds + "4.weight"), m->P(ds + "4.bias"), c2, B,
                             512, 4, 16, st));
  // im2col matrices feed one GEMM each: in the split modes they are written as its operand planes directly (no fp32 matrix, no
  // operand-split pre-pass over 151 MB)
  auto im2col = [&](const float* in, float* col, int Hin, int Cin) -> int {
    const int rows_c = B * (Hin / 2) * (Hin / 2), K = 9 * Cin;
    if (ds2_split_mode()) {
      ds2_model::ActPlanes cp;
      cp.ld = round32i(K);
      cp.hi = reinterpret_cast<unsigned short*>(m->alloc_bytes((size_t)rows_c * cp.ld * 2));
      cp.lo = reinterpret_cast<unsigned short*>(m->alloc_bytes((size_t)rows_c * cp.ld * 2));
      if (!cp.hi || !cp.lo) { ds2_set_error("memory_encoder: workspace exhausted (im2col planes)"); return DS2_ERR_STATE; }
      m->act_planes[col] = cp;
      return launch_im2col3x3s2_split(in, cp.hi, cp.lo, cp.ld, B, Hin, Cin, st);
    }
    return launch_im2col3x3s2(in, col, B, Hin, Cin, st);
  };
  ALLOC(col3, (size_t)B * 16384 * 144);
  TRY(im2col(c2, col3, 256, 16));
  ALLOC(g3, (size_t)B * 16384 * 64);
  TRY(gemm(st, B * 16384, 64, 144, col3, 144, m->P("@mds6_w"), 144, m->P(ds + "6.bias"), g3, 64, DS2_ACT_NONE, nullptr, 0, 0, nullptr, true, m));
  ALLOC(c3, (size_t)B * 16384 * 64);
  TRY(layernorm(m, st, ds + "7", g3, c3, B * 16384, 64, 1e-6f, DS2_ACT_GELU));
  ALLOC(col4, (size_t)rows * 576);
  TRY(im2col(c3, col4, 128, 64));
  ALLOC(g4, (size_t)rows * 256);
  TRY(gemm(st, rows, 256, 576, col4, 576, m->P("@mds9_w"), 576, m->P(ds + "9.bias"), g4, 256, DS2_ACT_NONE, nullptr, 0, 0, nullptr, true, m));
  ALLOC(c4, (size_t)rows * 256);
  TRY(layernorm(m, st, ds + "10", g4, c4, rows, 256, 1e-6f, DS2_ACT_GELU, true));   // consumer: encoder.12 GEMM
  // x = pix_feat_proj(pix_feat) + mask_downsampler(masks)   (memory_encoder.py:172-175); pix_feat is shared by all objects
  const int pf_rows = pix_shared ? TOK : rows;
  ALLOC(pf, (size_t)pf_rows * 256);
  TRY(linear(m, st, me + ".pix_feat_proj", pf_rows, 256, 256, fpn2, 256, pf, 256));
  ALLOC(x, (size_t)rows * 256);
  TRY(linear(m, st, ds + "12", rows, 256, 256, c4, 256, x, 256, DS2_ACT_NONE, pf, 256, pix_shared ? TOK : 0));
  // Fuser: 2 x CXBlock (memory_encoder.py:104-117)
  ALLOC(d, (size_t)rows * 256);
  ALLOC(t, (size_t)rows * 256);
  ALLOC(h, (size_t)rows * 1024);
  for (int l = 0; l < 2; ++l) {
    const std::string p = me + ".fuser.layers." + std::to_string(l);
    TRY(launch_dwconv7(x, m->P("@dw_w." + std::to_string(l)), m->P(p + ".dwconv.bias"), d, B, 64, 256, st));
    // the CXBlock's LayerNorm in the fused MLP's prologue when that kernel takes the block in its two-fp16-term form (else as its
    // own pass; same bits either way - verified bit for bit in round 5)
    MlpArgs probe{};
    probe.rows = rows; probe.D = 256; probe.H = 1024; probe.ldx = 256; probe.ldw1 = 256; probe.ldw2 = 1024; probe.ldo = 256; probe.ldr = 256;
    const bool ln_in = ds2_split_mode() && f16x2_enabled() && mlp256_supported(probe) &&
                       m->P(p + ".norm.weight") && m->P(p + ".norm.bias");
    const LnFuse lni{m->P(p + ".norm.weight"), m->P(p + ".norm.bias"), 1e-6f, nullptr, nullptr};
    if (!ln_in) TRY(layernorm(m, st, p + ".norm", d, t, rows, 256, 1e-6f, DS2_ACT_NONE, true));
    TRY(mlp2(m, st, p + ".pwconv1", p + ".pwconv2", rows, 1024, ln_in ? d : t, h, x, DS2_ACT_GELU, x, m->P(p + ".gamma"),
             l == 1, f16x2_enabled(), nullptr, nullptr, ln_in ? &lni : nullptr));   // the last block's result feeds out_proj
  }
  if (out_f32) {      // MemoryEncoder.forward's own output (no no_obj_embed_spatial, no bf16 storage rounding)
    TRY(linear(m, st, me + ".out_proj", rows, 64, 256, x, 256, out_f32, 64));
  } else {
    ALLOC(o, (size_t)rows * 64);
    TRY(linear(m, st, me + ".out_proj", rows, 64, 256, x, 256, o, 64));
    TRY(launch_memfeat_finish(o, obj_logits, m->P("no_obj_embed_spatial"), maskmem_bf16, B, TOK, 64, st));
  }
  CHECK_PARAMS();
  return DS2_OK;
}

// ------------------------------------------------------------------------------------------------ F3 (mask prompts)
extern "C" int ds2_resize_aa(const float* in, int32_t B, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, float in_scale,
                             float in_bias, float threshold, float* work, float* out, void* stream) {
  DS2_REQUIRE(in && work && out && B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "ds2_resize_aa: bad argument");
  return launch_resize_aa(in, work, out, B, Hin, Win, Hout, Wout, in_scale, in_bias, threshold, (hipStream_t)stream);
}

extern "C" int ds2_mask_prompt_prepare(ds2_model* m, int32_t B, const float* mask, float* mask_ds, float* obj_logits,
                                       int32_t* work, void* stream) {
  DS2_REQUIRE(m && m->finalized && B > 0 && mask && mask_ds && obj_logits && work, "ds2_mask_prompt_prepare: bad argument");
  ModelScope _dg(m);
  const float* w = m->P("mask_downsample.weight");
  const float* b = m->P("mask_downsample.bias");
  CHECK_PARAMS();
  return launch_mask_downsample4(mask, w, b, mask_ds, work, obj_logits, B, m->cfg.image_size, (hipStream_t)stream);
}

extern "C" int ds2_obj_ptr_gate(ds2_model* m, int32_t B, float* obj_ptr, const float* obj_logits, void* stream) {
  DS2_REQUIRE(m && m->finalized && B > 0 && obj_ptr && obj_logits, "ds2_obj_ptr_gate: bad argument");
  ModelScope _dg(m);
  const float* no = m->P("no_obj_ptr");
  CHECK_PARAMS();
  return launch_ptr_gate(obj_ptr, obj_logits, no, B, 256, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ A15
extern "C" int ds2_mask_output(ds2_model* m, const float* low_res, int32_t B, int32_t Hv, int32_t Wv, float* logits,
                               uint8_t* packed, void* stream) {
  DS2_REQUIRE(m && low_res && B > 0 && Hv > 0 && Wv > 0 && (logits || packed), "ds2_mask_output: bad argument");
  return launch_mask_output(low_res, B, 256, Hv, Wv, logits, packed, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ primitives
extern "C" int ds2_op_gemm(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* W, int32_t ldw,
                           const float* bias, float* C, int32_t ldc, int32_t act, const float* gamma, const float* R,
                           int32_t ldr, int32_t r_mod, void* stream) {
  return gemm((hipStream_t)stream, M, N, K, A, lda, W, ldw, bias, C, ldc, act, R, ldr, r_mod, gamma);
}
// C = act(A W^T + bias) returned as the consumer GEMM's operand planes (bf16 hi / lo, [M, round32(N)]) - what the Hiera MLP's first
// Linear layer writes (GemmSplitArgs::C_hi / C_lo); split modes only
extern "C" int ds2_op_gemm_planes(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* W, int32_t ldw,
                                  const float* bias, int32_t act, uint16_t* out_hi, uint16_t* out_lo, void* stream) {
  DS2_REQUIRE(ds2_split_mode() && A && W && out_hi && out_lo && M > 0 && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0,
              "ds2_op_gemm_planes: bad argument (split modes only)");
  hipStream_t st = (hipStream_t)stream;
  const int Kp = round32i(K), ldcp = round32i(N);
  const size_t ab = (size_t)M * Kp * 2, wb = (size_t)N * Kp * 2;
  TRY(g_gemm_ctx.require(2 * ab + 2 * wb + 1024, st));
  unsigned short* ah = reinterpret_cast<unsigned short*>(g_gemm_ctx.scratch);
  unsigned short* al = ah + (size_t)M * Kp;
  unsigned short* wh = al + (size_t)M * Kp;
  unsigned short* wl = wh + (size_t)N * Kp;
  // mode bf16x3k at a shape the MX form takes (what mlp.layers.0 of Hiera stages 3 / 4 is at hiera_l): the product is the
  // two-MFMA-equivalent one and the planes are MX ACTIVATION planes (common.h) - what that layer hands to mlp.layers.1
  if (gemm_mx_wanted(M, N, K, act, false, 0, false, false, true, false, bias != nullptr, false) && (reinterpret_cast<uintptr_t>(bias) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out_hi) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_lo) & 15) == 0) {
    TRY(launch_split_rows(A, lda, M, K, ah, al, Kp, st, false, DS2_PLANES_MX_A));
    TRY(launch_split_rows(W, ldw, N, K, wh, wl, Kp, st, false, DS2_PLANES_MX_W));
    GemmSplitArgs g{};
    g.M = M; g.N = N; g.Kp = Kp; g.A_hi = ah; g.A_lo = al; g.lda = Kp; g.W_hi = wh; g.W_lo = wl; g.ldw = Kp;
    g.bias = bias; g.act = act; g.C_hi = out_hi; g.C_lo = out_lo; g.ldcp = ldcp; g.mx = 1; g.c_mx = 1;
    return launch_gemm_split(g, st);
  }
  TRY(launch_split_rows(A, lda, M, K, ah, al, Kp, st));
  TRY(launch_split_rows(W, ldw, N, K, wh, wl, Kp, st));
  if (ldcp != N) {
    DS2_CHECK_HIP(hipMemsetAsync(out_hi, 0, (size_t)M * ldcp * 2, st));
    DS2_CHECK_HIP(hipMemsetAsync(out_lo, 0, (size_t)M * ldcp * 2, st));
  }
  GemmSplitArgs g{};
  g.M = M; g.N = N; g.Kp = Kp; g.A_hi = ah; g.A_lo = al; g.lda = Kp; g.W_hi = wh; g.W_lo = wl; g.ldw = Kp;
  g.bias = bias; g.act = act; g.C_hi = out_hi; g.C_lo = out_lo; g.ldcp = ldcp;
  return launch_gemm_split(g, st);
}
// x [rows, cols] fp32 -> its two operand planes [rows, round-up-32(cols)] in one of the formats of common.h: 0 bf16 hi / lo, 1 MX
// activation, 2 MX weight, 3 fp16 hi / lo (test hook of the plane formats)
extern "C" int ds2_op_split_planes(const float* x, int32_t ldx, int32_t rows, int32_t cols, int32_t fmt, uint16_t* p1, uint16_t* p2, void* stream) {
  DS2_REQUIRE(x && p1 && p2 && rows > 0 && cols > 0 && cols % 4 == 0 && ldx % 4 == 0 && fmt >= 0 && fmt <= 3, "ds2_op_split_planes: bad argument");
  return launch_split_rows(x, ldx, rows, cols, p1, p2, round32i(cols), (hipStream_t)stream, fmt == 3, fmt == 3 ? 0 : fmt);
}
extern "C" int ds2_op_linear_small(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* W, int32_t ldw,
                                   const float* bias, float* C, int32_t ldc, int32_t act, const float* gamma, const float* R,
                                   int32_t ldr, int32_t r_mod, void* stream) {
  DS2_REQUIRE(M > 0 && M <= 128 && N > 0 && K > 0 && A && W && C, "ds2_op_linear_small: bad argument (1 <= M <= 128)");
  hipStream_t st = (hipStream_t)stream;
  TRY(g_gemm_ctx.require((size_t)N * K * 4, st));           // the model path caches the transposed weight; a primitive call redoes it
  float* wt = reinterpret_cast<float*>(g_gemm_ctx.scratch);
  TRY(launch_transpose_w(W, ldw, N, K, wt, st));
  SkinnyArgs g{M, N, K, A, lda, wt, bias, gamma, R, ldr, r_mod, C, ldc, act};
  return launch_skinny_linear(g, st);
}
extern "C" int ds2_op_mlp(int32_t rows, int32_t H, const float* X, const float* W1, const float* b1, const float* W2, const float* b2,
                          const float* gamma, const float* R, float* out, int32_t act, void* stream) {
  DS2_REQUIRE(rows > 0 && H > 0 && X && W1 && W2 && out, "ds2_op_mlp: bad argument");
  // (DS2_OP_MLP_F16X2=1: the two-term fp16 form the memory attention / memory encoder use in mode bf16x3k - tests/test_hip_ops.py)
  const char* e2 = getenv("DS2_OP_MLP_F16X2");
  const int rc = mlp_fused(nullptr, g_gemm_ctx, (hipStream_t)stream, rows, H, X, 256, W1, b1, W2, b2, gamma, R, 256, out, 256, act, false,
                           e2 && atoi(e2) != 0);
  DS2_REQUIRE(rc != DS2_ERR_UNSUPPORTED, "ds2_op_mlp: needs a bf16x3 mode, width 256, H a multiple of 64 (<= 4096), act none / relu / gelu");
  return rc;
}
extern "C" int ds2_op_layernorm(const float* x, const float* w, const float* b, float* y, int32_t rows, int32_t C, float eps,
                                int32_t act, void* stream) {
  return launch_layernorm(x, C, w, b, y, C, rows, C, eps, act, (hipStream_t)stream);
}
extern "C" int ds2_op_attention(const float* q, const float* k, const float* v, float* o, int32_t ldq, int32_t ldk, int32_t ldv,
                                int32_t ldo, int32_t batch, int32_t heads, int32_t D, int32_t DV, int32_t Lq, int32_t Lk,
                                float scale, int32_t win_q, int32_t win_k, int32_t Hq, int32_t Wq, int32_t Hk, int32_t Wk,
                                int32_t nwx, const float* k_pad, const float* v_pad, void* stream) {
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = o;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.batch = batch; a.heads = heads; a.D = D; a.DV = DV; a.Lq = Lq; a.Lk = Lk; a.scale = scale;
  a.win_q = win_q; a.win_k = win_k; a.Hq = Hq; a.Wq = Wq; a.Hk = Hk; a.Wk = Wk; a.nwx = nwx;
  a.k_pad = k_pad; a.v_pad = v_pad;
  if (ds2_split_mode() && attention_hg_supported(a)) {   // as the image encoder does for its global-attention blocks
    hipStream_t st = (hipStream_t)stream;
    const size_t kb = (attention_hg_k_bytes(a) + 255) & ~(size_t)255, vb = attention_hg_vt_bytes(a);
    TRY(g_gemm_ctx.require(kb + vb, st));
    return launch_attention_hg(a, g_gemm_ctx.scratch, g_gemm_ctx.scratch + kb, st);
  }
  return launch_attention(a, (hipStream_t)stream);
}
