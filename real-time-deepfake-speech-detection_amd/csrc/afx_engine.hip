// libafx engine: weight store + forward orchestration + the C ABI of include/afx.h.
// Host code (C++) over the HIP runtime; the only device code here is a few one-off
// weight-preparation kernels.  The forward is a fixed sequence of asynchronous kernel
// launches on the caller's stream: no allocation, no synchronisation, no host<->device
// copies inside afx_forward (it can be captured into a hipGraph by the caller).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/afx.h"
#include "afx_common.h"
#include "afx_kernels.h"
#include "afx_aasist.h"

using namespace afx;

// ---------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}
#define HIP_OK(expr)                                                               \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_));     \
  } while (0)
#define KOK(expr)                                    \
  do {                                               \
    const char* m_ = (expr);                         \
    if (m_) return fail("%s", m_);                   \
  } while (0)

// ---------------------------------------------------------------------------------
// model constants (XLS-R 300M trunk)
// ---------------------------------------------------------------------------------
static const int kConvK[7] = {10, 3, 3, 3, 3, 2, 2};
static const int kConvS[7] = {5, 2, 2, 2, 2, 2, 2};
constexpr int kC = 512;      // conv channels
constexpr int kD = 1024;     // encoder width
constexpr int kF = 4096;     // FFN width
constexpr int kH = 16;       // heads
constexpr int kPosK = 128;   // positional conv taps
constexpr int kPosG = 16;    // positional conv groups
constexpr int kPosPad = 64;  // = kPosK / 2
constexpr float kLnEps = 1e-5f;
constexpr float kBnEps = 1e-5f;

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static void conv_lengths(int L, int* T) {
  for (int i = 0; i < 7; ++i) {
    L = L >= kConvK[i] ? (L - kConvK[i]) / kConvS[i] + 1 : 0;
    T[i] = L;
  }
}

// one-off device helpers ------------------------------------------------------------
__global__ void bn_fold_kernel(const float* w, const float* b, const float* m, const float* v, float eps, int n,
                               float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float s = w[i] / sqrtf(v[i] + eps);
    scale[i] = s;
    shift[i] = b[i] - m[i] * s;
  }
}

__global__ void absmax_kernel(const float* x, int n, float* out) {  // out[0] = max |x| (one workgroup)
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(x[i]));
  __shared__ float red[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

struct FT {  // fp32 device tensor owned by the engine
  float* p = nullptr;
  size_t n = 0;
  std::vector<int64_t> shape;
};

struct TapRec {
  float* p = nullptr;  // engine-owned fp32 copy
  size_t cap = 0, n = 0;
};

struct ConfBlock {
  void *ff1_w1, *ff1_w2, *ff2_w1, *ff2_w2, *wqkv, *wout, *pw1, *pw2;
  float *bn_scale, *bn_shift;
  void* rel_h;          // rel_pos_emb in operand type, rows padded to 64 (matrix-core attention)
  float* chain_prm[3];  // per fused chain: the per-column vectors packed into one 8-KB block (ConfChainArgs::params)
};

struct Profiler;  // per-engine launch timing (below)

struct afx_engine {
  afx_config cfg;
  bool s3 = false;  // split precision (AFX_DT_FP16X3): dt == DT_FP32 for every kernel but the dense products
  Profiler* prof = nullptr;  // owned; no process-wide map: two engines on two host threads share nothing
  int dt;
  size_t hsz;  // bytes per operand element
  std::vector<void*> allocs;
  std::unordered_map<std::string, FT> f;  // raw fp32 tensors by canonical name
  std::unordered_set<std::string> loaded;
  bool finalized = false;
  bool taps_on = false;
  // A/B switches between forms of the same op (afx_engine_set; per engine: another engine of the process is not touched)
  int posconv_sliding = 1;  // positional conv: sliding-window kernel (0: chunked-K GEMM)
  int conf_attn_mfma = 1;   // Conformer attention on the matrix cores (0: the fp32 VALU kernel)
  int fuse_conformer = 1;   // Conformer block: row-local chains fused (afx_conformer_fused.hip); 0 = per-op path
  int fuse_conv_ln = 1;     // conv layers 1-6: LayerNorm + GELU in the GEMM epilogue (0: two kernels)
  int gemm_small_deep = 1;  // products with at most two 128x64 tiles per CU: the deep form of that tile (0: the two-buffer form)
  std::unordered_map<std::string, TapRec> taps;
  // overflow guard: device counters [0] rows of the trunk's final LayerNorm with non-finite statistics, [1] non-finite logits
  // (written by those kernels of every forward, read and cleared by afx_check_finite)
  int* nonfinite = nullptr;
  // split precision: the scale each LayerNorm's output takes as a pair-form operand, keyed by the LayerNorm's gamma pointer and
  // chosen at finalize from |y| <= sqrt(C) max|gamma| + max|beta|: the largest power of two <= kS3ScaleBounded that keeps the hi
  // half inside fp16 -- a LayerNorm output cannot overflow whatever the checkpoint's gains are
  std::unordered_map<const float*, float> ln_scale;

  // trunk, packed operand-type weights
  void* convw[7] = {nullptr};
  void* conv0pack = nullptr;  // layer 0 as the split-precision fp16 MFMA operand (layer_norm mode, half-precision engines)
  void* projw = nullptr;
  void* posw = nullptr;
  float* pos_norm = nullptr;
  std::vector<void*> wqkv, wo, w1, w2;
  std::vector<float*> bqkv;
  // Conformer head
  int E = 0, Ep = 0, heads = 0, dh = 0, inner = 0, FF = 0, FFp = 0, C2 = 0, C2p = 0, ck = 0, nblk = 0;
  void* conf_ll = nullptr;
  float conf_bn_scale = 1.f, conf_bn_shift = 0.f;
  std::vector<ConfBlock> blk;
  // AASIST head
  AasistWeights aw;

  // a packed weight matrix [rows][k] (+, in split precision, its per-row scales behind it: wscale())
  void* walloc(size_t rows, size_t k) { return dalloc(rows * k * hsz + (s3 ? rows * 4 : 0)); }
  float* wscale(const void* w, size_t rows, size_t k) const { return s3 ? (float*)((char*)w + rows * k * 4) : nullptr; }
  void* dalloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    allocs.push_back(p);
    return p;
  }
  const float* F(const std::string& k) const {
    auto it = f.find(k);
    return it == f.end() ? nullptr : it->second.p;
  }
};

// ---------------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------------
static void prof_forget(afx_engine* e);
extern "C" const char* afx_last_error(void) { return g_err; }
extern "C" const char afx_build_id_str[];  // build_id.gen.c (Makefile): hash of the sources this library was built from
extern "C" const char* afx_build_id(void) { return afx_build_id_str; }
extern "C" const char* afx_version(void) {
  static char v[96] = "";
  if (!v[0]) snprintf(v, sizeof v, "afx 0.4 (gfx950) build %s hip %d.%d.%d", afx_build_id_str, HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH);
  return v;
}
// The HIP version the library's code objects and launch stubs were compiled against, and the version of the runtime this
// process actually resolved (a torch wheel bundles its own libamdhip64; afx/_lib.py loads torch first so that the process has
// ONE runtime): the host side refuses a different major version and warns when the runtime is older than the toolchain.
extern "C" int afx_hip_versions(int* build, int* runtime) {
  if (build) *build = HIP_VERSION;
  int rt = 0;
  if (hipRuntimeGetVersion(&rt) != hipSuccess) return fail("afx_hip_versions: hipRuntimeGetVersion failed");
  if (runtime) *runtime = rt;
  return 0;
}

extern "C" int afx_create(const afx_config* cfg, afx_handle* out) {
  if (!cfg || !out) return fail("afx_create: null argument");
  const bool head_only = cfg->arch == AFX_ARCH_CONFORMER_HEAD;  // MyConformer alone: no trunk in this handle
  if (!head_only && (cfg->n_layers < 1 || cfg->n_layers > 24))
    return fail("Number of layers must be at least 1 and at most 24.");  // models/fe.py:60-62
  if (cfg->dtype != AFX_DT_BF16 && cfg->dtype != AFX_DT_FP16 && cfg->dtype != AFX_DT_FP32 && cfg->dtype != AFX_DT_FP16X3)
    return fail("afx_create: unknown dtype %d", cfg->dtype);
  if (cfg->arch < AFX_ARCH_SSL || cfg->arch > AFX_ARCH_CONFORMER_HEAD) return fail("afx_create: unknown arch %d", cfg->arch);
  if (cfg->extractor_mode != AFX_EXTRACTOR_LAYER_NORM && cfg->extractor_mode != AFX_EXTRACTOR_GROUP_NORM)
    return fail("afx_create: unknown extractor_mode %d", cfg->extractor_mode);
  if (cfg->extractor_mode == AFX_EXTRACTOR_GROUP_NORM && cfg->pre_emphasis)
    return fail("afx_create: fused pre-emphasis is built for the layer_norm extractor only");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail("afx_create: no HIP device visible -- this library has no CPU fallback");
  afx_engine* e = new afx_engine();
  e->cfg = *cfg;
  // split precision: activations and every non-GEMM kernel as in exact mode (fp32); the dense products on the fp16 matrix
  // pipe with hi / lo operand pairs (P_gemm below).  A weight element is 4 bytes either way (fp32, or an fp16 hi + lo pair).
  e->s3 = cfg->dtype == AFX_DT_FP16X3;
  e->dt = e->s3 ? (int)DT_FP32 : cfg->dtype;
  e->hsz = dtype_size(e->dt);
  const int nl = head_only ? 0 : cfg->n_layers;
  e->cfg.n_layers = nl;
  e->wqkv.assign(nl, nullptr);
  e->wo.assign(nl, nullptr);
  e->w1.assign(nl, nullptr);
  e->w2.assign(nl, nullptr);
  e->bqkv.assign(nl, nullptr);
  bool ok = true;
  ok &= (e->nonfinite = (int*)e->dalloc(16)) != nullptr && hipMemset(e->nonfinite, 0, 16) == hipSuccess;
  if (!head_only) {
    for (int i = 1; i < 7; ++i) ok &= (e->convw[i] = e->walloc(kC, (size_t)kC * kConvK[i])) != nullptr;
    ok &= (e->projw = e->walloc(kD, kC)) != nullptr;
    ok &= (e->posw = e->walloc(kD, (size_t)(kD / kPosG) * kPosK)) != nullptr;
    ok &= (e->pos_norm = (float*)e->dalloc(kPosK * 4)) != nullptr;
  }
  for (int l = 0; l < nl && ok; ++l) {
    ok &= (e->wqkv[l] = e->walloc(3 * kD, kD)) != nullptr;
    ok &= (e->wo[l] = e->walloc(kD, kD)) != nullptr;
    ok &= (e->w1[l] = e->walloc(kF, kD)) != nullptr;
    ok &= (e->w2[l] = e->walloc(kD, kF)) != nullptr;
    ok &= (e->bqkv[l] = (float*)e->dalloc((size_t)3 * kD * 4)) != nullptr;
  }
  if (cfg->arch == AFX_ARCH_CONFORMER || head_only) {
    e->E = cfg->conf_emb;
    e->heads = cfg->conf_heads;
    e->ck = cfg->conf_kernel;
    e->nblk = cfg->conf_blocks;
    if (e->E <= 0 || e->heads <= 0 || e->E % e->heads || e->E % 4 || e->ck <= 0 || e->nblk <= 0) {
      afx_destroy(e);  // (the message reads the caller's cfg: `e` is gone -- a use-after-free found by `make asan`)
      return fail("afx_create: bad Conformer configuration (emb %d heads %d kernel %d blocks %d)", cfg->conf_emb, cfg->conf_heads,
                  cfg->conf_kernel, cfg->conf_blocks);
    }
    e->dh = e->E / e->heads;
    e->inner = e->dh * e->heads;
    e->Ep = round_up(e->E, 64);
    e->FF = 4 * e->E;
    e->FFp = round_up(e->FF, 64);
    e->C2 = 2 * e->E;
    e->C2p = round_up(e->C2, 64);
    if (!head_only) ok &= (e->conf_ll = e->walloc(e->E, kD)) != nullptr;
    e->blk.resize(e->nblk);
    for (int b = 0; b < e->nblk && ok; ++b) {
      ConfBlock& B = e->blk[b];
      ok &= (B.ff1_w1 = e->walloc(e->FF, e->Ep)) != nullptr;
      ok &= (B.ff1_w2 = e->walloc(e->E, e->FFp)) != nullptr;
      ok &= (B.ff2_w1 = e->walloc(e->FF, e->Ep)) != nullptr;
      ok &= (B.ff2_w2 = e->walloc(e->E, e->FFp)) != nullptr;
      ok &= (B.wqkv = e->walloc(3 * e->inner, e->Ep)) != nullptr;
      ok &= (B.wout = e->walloc(e->E, e->Ep)) != nullptr;
      ok &= (B.pw1 = e->walloc(2 * e->C2, e->Ep)) != nullptr;
      ok &= (B.pw2 = e->walloc(e->E, e->C2p)) != nullptr;
      ok &= (B.bn_scale = (float*)e->dalloc((size_t)e->C2 * 4)) != nullptr;
      ok &= (B.bn_shift = (float*)e->dalloc((size_t)e->C2 * 4)) != nullptr;
      for (int st = 0; st < 3; ++st) ok &= (B.chain_prm[st] = (float*)e->dalloc((kChainParamFloats + kChainScaleFloats) * 4)) != nullptr;
      ok &= (B.rel_h = e->dalloc((size_t)1025 * 64 * e->hsz)) != nullptr;
    }
  }
  if (!ok) {
    afx_destroy(e);
    return fail("afx_create: device allocation failed");
  }
  *out = e;
  return 0;
}

extern "C" void afx_destroy(afx_handle h) {
  if (!h) return;
  prof_forget(h);
  for (void* p : h->allocs) (void)hipFree(p);
  for (auto& t : h->taps)
    if (t.second.p) (void)hipFree(t.second.p);
  delete h;
}

extern "C" int afx_enable_taps(afx_handle h, int on) {
  if (!h) return fail("afx_enable_taps: null handle");
  h->taps_on = on != 0;
  return 0;
}

// Overflow guard.  The half-precision modes keep operand COPIES in fp16 / bf16 (LayerNorm outputs, q | k | v rows, attention
// output, the GELU'd FFN hidden, conv-stack activations; split precision: fp16 hi / lo pairs of scale x the value).  A
// trained checkpoint with outlier channels can push one of them past the format's range (fp16: 65 504; fp16x3: 65 504 /
// its activation scale: kS3ScaleFree = 1 for unbounded operands); the inf then turns every later row statistic into NaN.  Rather than hand back NaN -- or, behind the AASIST
// head's max-pooling and top-k, finite garbage -- scores, the trunk's final LayerNorm and the kernels that write the logits
// count what is not finite (no cost: one compare per row / logit), and this call turns a non-zero count into an error:
// it waits for `stream`, reads and clears the counters.  Scoring loops call it once, before they write a score.
extern "C" int afx_check_finite(afx_handle h, void* stream) {
  if (!h) return fail("afx_check_finite: null handle");
  int c[2] = {0, 0};
  hipStream_t s = (hipStream_t)stream;
  HIP_OK(hipMemcpyAsync(c, h->nonfinite, 8, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  if (c[0] || c[1]) {
    HIP_OK(hipMemsetAsync(h->nonfinite, 0, 8, s));
    return fail("non-finite values since the last check: %d feature rows of the trunk's final LayerNorm, %d logits -- an operand copy left the "
                "range of this engine's precision (%s); the scores of these batches are invalid.  Use dtype \"fp32\" (exact mode) for this checkpoint",
                c[0], c[1], h->s3 ? "fp16x3: fp16 hi / lo pairs" : (h->dt == AFX_DT_FP16 ? "fp16 operands: |x| <= 65504" : (h->dt == AFX_DT_BF16 ? "bf16 operands" : "fp32")));
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------
static bool starts_with(const std::string& s, const char* p) { return s.compare(0, strlen(p), p) == 0; }
static bool ends_with(const std::string& s, const char* p) {
  const size_t n = strlen(p);
  return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}
static size_t numel(const int64_t* shape, int ndim) {
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  return n;
}
static int expect_shape(const char* name, const int64_t* shape, int ndim, std::initializer_list<int64_t> want) {
  size_t have = numel(shape, ndim), w = 1;
  for (int64_t v : want) w *= (size_t)v;
  if (have != w) return fail("afx_load_weight: %s has %zu elements, expected %zu", name, have, w);
  return 0;
}

static int store_raw(afx_engine* e, const std::string& key, const float* src, const int64_t* shape, int ndim,
                     hipStream_t s) {
  FT& t = e->f[key];
  const size_t n = numel(shape, ndim);
  if (!t.p || t.n != n) {
    if (t.p) {  // reloaded with another size: release the old buffer now, not at destroy
      for (auto it = e->allocs.begin(); it != e->allocs.end(); ++it)
        if (*it == t.p) { e->allocs.erase(it); break; }
      (void)hipFree(t.p);
    }
    t.p = (float*)e->dalloc(n * 4);
    if (!t.p) return fail("afx_load_weight: device allocation failed for %s", key.c_str());
    t.n = n;
  }
  t.shape.assign(shape, shape + ndim);
  HIP_OK(hipMemcpyAsync(t.p, src, n * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

// pack a weight into the engine's operand form; split precision: fp32 rows first, then hi / lo pairs + row scales in place
static const char* pack_linear(afx_engine* e, const float* src, int N, int K, int Kpad, void* dst, float* scale, hipStream_t s) {
  if (const char* m = launch_pack_linear(src, N, K, Kpad, dst, e->dt, s)) return m;
  return e->s3 ? launch_split_weight_rows(dst, N, Kpad, scale, s) : nullptr;
}

// trunk tensor (name without the "ssl_model.model." prefix)
static int load_ssl(afx_engine* e, const std::string& k, const float* src, const int64_t* shape, int ndim,
                    hipStream_t s) {
  int i = 0, n = 0;
  char tail[96];
  if (sscanf(k.c_str(), "feature_extractor.conv_layers.%d.%95s", &i, tail) == 2 && !strcmp(tail, "0.weight")) {
    if (i < 0 || i > 6) return fail("afx_load_weight: conv layer %d out of range", i);
    if (i == 0) {
      if (expect_shape(k.c_str(), shape, ndim, {kC, 1, kConvK[0]})) return 1;
      return store_raw(e, "ssl." + k, src, shape, ndim, s);
    }
    if (expect_shape(k.c_str(), shape, ndim, {kC, kC, kConvK[i]})) return 1;
    KOK(launch_pack_conv(src, kC, kC, kConvK[i], e->convw[i], e->dt, s));
    if (e->s3) KOK(launch_split_weight_rows(e->convw[i], kC, kC * kConvK[i], e->wscale(e->convw[i], kC, (size_t)kC * kConvK[i]), s));
    return 0;
  }
  if (k == "post_extract_proj.weight") {
    if (expect_shape(k.c_str(), shape, ndim, {kD, kC})) return 1;
    KOK(pack_linear(e, src, kD, kC, kC, e->projw, e->wscale(e->projw, kD, kC), s));
    return 0;
  }
  if (sscanf(k.c_str(), "encoder.layers.%d.%95s", &n, tail) == 2) {
    if (n < 0) return fail("afx_load_weight: bad layer index in %s", k.c_str());
    if (n >= e->cfg.n_layers) return 0;  // a deeper checkpoint than this (truncated) trunk keeps: ignore
    const std::string t = tail;
    static const char* proj[3] = {"self_attn.q_proj.", "self_attn.k_proj.", "self_attn.v_proj."};
    for (int j = 0; j < 3; ++j) {
      if (t == std::string(proj[j]) + "weight") {
        if (expect_shape(k.c_str(), shape, ndim, {kD, kD})) return 1;
        KOK(pack_linear(e, src, kD, kD, kD, (char*)e->wqkv[n] + (size_t)j * kD * kD * e->hsz,
                        e->s3 ? e->wscale(e->wqkv[n], 3 * kD, kD) + j * kD : nullptr, s));
        return 0;
      }
      if (t == std::string(proj[j]) + "bias") {
        if (expect_shape(k.c_str(), shape, ndim, {kD})) return 1;
        HIP_OK(hipMemcpyAsync(e->bqkv[n] + j * kD, src, kD * 4, hipMemcpyDeviceToDevice, s));
        return 0;
      }
    }
    if (t == "self_attn.out_proj.weight") {
      if (expect_shape(k.c_str(), shape, ndim, {kD, kD})) return 1;
      KOK(pack_linear(e, src, kD, kD, kD, e->wo[n], e->wscale(e->wo[n], kD, kD), s));
      return 0;
    }
    if (t == "fc1.weight") {
      if (expect_shape(k.c_str(), shape, ndim, {kF, kD})) return 1;
      KOK(pack_linear(e, src, kF, kD, kD, e->w1[n], e->wscale(e->w1[n], kF, kD), s));
      return 0;
    }
    if (t == "fc2.weight") {
      if (expect_shape(k.c_str(), shape, ndim, {kD, kF})) return 1;
      KOK(pack_linear(e, src, kD, kF, kF, e->w2[n], e->wscale(e->w2[n], kD, kF), s));
      return 0;
    }
    return store_raw(e, "ssl." + k, src, shape, ndim, s);
  }
  // everything else on the path is a small fp32 tensor; off-path keys are dropped
  if (starts_with(k, "quantizer.") || starts_with(k, "project_q.") || starts_with(k, "final_proj.") ||
      k == "mask_emb" || starts_with(k, "target_glu") || starts_with(k, "layer_norm_") )
    return 0;
  return store_raw(e, "ssl." + k, src, shape, ndim, s);
}

static int load_conformer(afx_engine* e, const std::string& k, const float* src, const int64_t* shape, int ndim,
                          hipStream_t s) {
  const int E = e->E, Ep = e->Ep;
  if (k == "LL.weight") {
    if (expect_shape(k.c_str(), shape, ndim, {E, kD})) return 1;
    KOK(pack_linear(e, src, E, kD, kD, e->conf_ll, e->wscale(e->conf_ll, E, kD), s));
    return 0;
  }
  int b = 0;
  char tail[96];
  if (sscanf(k.c_str(), "conformer.encoder_blocks.%d.%95s", &b, tail) == 2) {
    if (b < 0 || b >= e->nblk) return fail("afx_load_weight: Conformer block %d out of range (n_encoders=%d)", b, e->nblk);
    ConfBlock& B = e->blk[b];
    const std::string t = tail;
    float* qkv_sc = e->wscale(B.wqkv, 3 * e->inner, Ep);
    struct { const char* name; void* dst; int N, K, Kp; float* sc; } lin[] = {
        {"ff1.fn.fn.net.0.weight", B.ff1_w1, e->FF, E, Ep, e->wscale(B.ff1_w1, e->FF, Ep)},
        {"ff1.fn.fn.net.3.weight", B.ff1_w2, E, e->FF, e->FFp, e->wscale(B.ff1_w2, E, e->FFp)},
        {"ff2.fn.fn.net.0.weight", B.ff2_w1, e->FF, E, Ep, e->wscale(B.ff2_w1, e->FF, Ep)},
        {"ff2.fn.fn.net.3.weight", B.ff2_w2, E, e->FF, e->FFp, e->wscale(B.ff2_w2, E, e->FFp)},
        {"attn.fn.to_q.weight", B.wqkv, e->inner, E, Ep, qkv_sc},
        {"attn.fn.to_kv.weight", (char*)B.wqkv + (size_t)e->inner * Ep * e->hsz, 2 * e->inner, E, Ep, qkv_sc ? qkv_sc + e->inner : nullptr},
        {"attn.fn.to_out.weight", B.wout, E, e->inner, Ep, e->wscale(B.wout, E, Ep)},
        {"conv.net.2.weight", B.pw1, 2 * e->C2, E, Ep, e->wscale(B.pw1, 2 * e->C2, Ep)},
        {"conv.net.7.weight", B.pw2, E, e->C2, e->C2p, e->wscale(B.pw2, E, e->C2p)},
    };
    for (auto& L : lin)
      if (t == L.name) {
        if (expect_shape(k.c_str(), shape, ndim, {L.N, L.K})) return 1;
        KOK(pack_linear(e, src, L.N, L.K, L.Kp, L.dst, L.sc, s));
        return 0;
      }
  }
  return store_raw(e, k, src, shape, ndim, s);
}

extern "C" int afx_load_weight(afx_handle h, const char* name, const float* dev_ptr, const int64_t* shape, int ndim,
                               void* stream) {
  if (!h || !name || !dev_ptr || (ndim > 0 && !shape)) return fail("afx_load_weight: null argument");
  hipStream_t s = (hipStream_t)stream;
  std::string k = name;
  if (starts_with(k, "module.")) k = k.substr(7);  // utils.py:13-43
  if (ends_with(k, "num_batches_tracked")) return 0;
  h->finalized = false;
  h->loaded.insert(k);
  if (starts_with(k, "ssl_model.model.")) return load_ssl(h, k.substr(16), dev_ptr, shape, ndim, s);
  if (h->cfg.arch == AFX_ARCH_SSL) {
    if (starts_with(k, "model.")) {
      h->loaded.insert("ssl_model." + k);
      return load_ssl(h, k.substr(6), dev_ptr, shape, ndim, s);
    }
    return 0;
  }
  if (h->cfg.arch == AFX_ARCH_CONFORMER_HEAD && (k == "LL.weight" || starts_with(k, "LL.") || starts_with(k, "first_bn."))) return 0;
  if (h->cfg.arch == AFX_ARCH_CONFORMER || h->cfg.arch == AFX_ARCH_CONFORMER_HEAD) return load_conformer(h, k, dev_ptr, shape, ndim, s);
  // AASIST head: everything is small fp32; bn1.* never influences the output (Q2)
  if (k.find(".bn1.") != std::string::npos) return 0;
  return store_raw(h, k, dev_ptr, shape, ndim, s);
}

static int need(afx_engine* e, const std::string& k) {
  if (!e->loaded.count(k)) return fail("afx_finalize: missing weight '%s'", k.c_str());
  return 0;
}

static int fold_bn(afx_engine* e, const std::string& p, int n, float* scale, float* shift, hipStream_t s) {
  const float *w = e->F(p + "weight"), *b = e->F(p + "bias"), *m = e->F(p + "running_mean"), *v = e->F(p + "running_var");
  if (!w || !b || !m || !v) return fail("afx_finalize: BatchNorm '%s*' incomplete", p.c_str());
  hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, b, m, v, kBnEps, n, scale, shift);
  HIP_OK(hipGetLastError());
  return 0;
}

// split precision: one scale per LayerNorm (afx_engine::ln_scale) from the maxima of its gain and offset
static int s3_layernorm_scales(afx_engine* h, hipStream_t s) {
  h->ln_scale.clear();
  if (!h->s3) return 0;
  std::vector<std::pair<const FT*, const FT*>> lns;
  for (const auto& kv : h->f) {
    const std::string& k = kv.first;
    if (kv.second.shape.size() != 1 || !ends_with(k, ".weight")) continue;
    auto b = h->f.find(k.substr(0, k.size() - 6) + "bias");
    if (b != h->f.end() && b->second.n == kv.second.n) lns.push_back({&kv.second, &b->second});
  }
  if (lns.empty()) return 0;
  float* d = nullptr;
  HIP_OK(hipMalloc((void**)&d, lns.size() * 8));
  for (size_t i = 0; i < lns.size(); ++i) {
    hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(256), 0, s, lns[i].first->p, (int)lns[i].first->n, d + 2 * i);
    hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(256), 0, s, lns[i].second->p, (int)lns[i].second->n, d + 2 * i + 1);
  }
  std::vector<float> m(2 * lns.size());
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(m.data(), d, m.size() * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (e != hipSuccess) return fail("afx_finalize: %s", hipGetErrorString(e));
  for (size_t i = 0; i < lns.size(); ++i) {
    const float bound = sqrtf((float)lns[i].first->n) * m[2 * i] + m[2 * i + 1];
    float sc = kS3ScaleBounded;
    while (sc * bound > 60000.f && sc > 1e-6f) sc *= 0.5f;  // (a non-finite gain stays non-finite: the overflow guard reports it)
    h->ln_scale[lns[i].first->p] = sc;
  }
  return 0;
}

extern "C" int afx_finalize(afx_handle h, void* stream) {
  if (!h) return fail("afx_finalize: null handle");
  hipStream_t s = (hipStream_t)stream;
  const std::string P = "ssl_model.model.";
  const bool gn = h->cfg.extractor_mode == AFX_EXTRACTOR_GROUP_NORM;
  const bool head_only = h->cfg.arch == AFX_ARCH_CONFORMER_HEAD;
  if (!head_only) {
    for (int i = 0; i < 7; ++i) {
      const std::string c = P + "feature_extractor.conv_layers." + std::to_string(i);
      if (need(h, c + ".0.weight")) return 1;
      if (gn) {  // wav2vec2-base: bias-free convs, GroupNorm(512,512) on layer 0 only
        if (i == 0 && (need(h, c + ".2.weight") || need(h, c + ".2.bias"))) return 1;
      } else if (need(h, c + ".0.bias") || need(h, c + ".2.1.weight") || need(h, c + ".2.1.bias")) {
        return 1;
      }
    }
    for (const char* k : {"layer_norm.weight", "layer_norm.bias", "post_extract_proj.weight", "post_extract_proj.bias",
                          "encoder.pos_conv.0.bias", "encoder.layer_norm.weight", "encoder.layer_norm.bias"})
      if (need(h, P + k)) return 1;
    for (int l = 0; l < h->cfg.n_layers; ++l) {
      const std::string L = P + "encoder.layers." + std::to_string(l) + ".";
      for (const char* k : {"self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight",
                            "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias",
                            "self_attn.out_proj.weight", "self_attn.out_proj.bias", "self_attn_layer_norm.weight",
                            "self_attn_layer_norm.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
                            "final_layer_norm.weight", "final_layer_norm.bias"})
        if (need(h, L + k)) return 1;
    }
    if (!gn && (h->dt != AFX_DT_FP32 || h->s3)) {  // conv layer 0 as the split-precision fp16 matrix-core operand
      if (!h->conv0pack && !(h->conv0pack = h->dalloc(conv0_pack_bytes()))) return fail("afx_finalize: device allocation failed");
      KOK(launch_conv0_pack(h->F("ssl.feature_extractor.conv_layers.0.0.weight"), h->F("ssl.feature_extractor.conv_layers.0.0.bias"),
                            h->conv0pack, s));
    }
    // positional conv: weight-norm (dim=2) folded into the packed operand
    const float* pv = h->F("ssl.encoder.pos_conv.0.weight_v");
    const float* pg = h->F("ssl.encoder.pos_conv.0.weight_g");
    const float* pw = h->F("ssl.encoder.pos_conv.0.weight");
    if (pv && pg) {
      KOK(launch_pack_posconv(pv, pg, kD, kD / kPosG, kPosK, h->pos_norm, h->posw, h->dt, s));
    } else if (pw) {
      KOK(launch_pack_posconv(pw, nullptr, kD, kD / kPosG, kPosK, h->pos_norm, h->posw, h->dt, s));
    } else {
      return fail("afx_finalize: missing weight '%sencoder.pos_conv.0.weight_g/weight_v'", P.c_str());
    }
    if (h->s3) KOK(launch_split_weight_rows(h->posw, kD, (kD / kPosG) * kPosK, h->wscale(h->posw, kD, (size_t)(kD / kPosG) * kPosK), s));
  }
  if (h->cfg.arch == AFX_ARCH_CONFORMER || head_only) {
    for (const char* k : {"LL.weight", "LL.bias", "first_bn.weight", "first_bn.bias", "first_bn.running_mean",
                          "first_bn.running_var", "conformer.class_token", "conformer.fc5.weight", "conformer.fc5.bias"})
      if (!(head_only && strncmp(k, "conformer.", 10)) && need(h, k)) return 1;
    for (int b = 0; b < h->nblk; ++b) {
      const std::string B = "conformer.encoder_blocks." + std::to_string(b) + ".";
      for (const char* k :
           {"ff1.fn.norm.weight", "ff1.fn.norm.bias", "ff1.fn.fn.net.0.weight", "ff1.fn.fn.net.0.bias",
            "ff1.fn.fn.net.3.weight", "ff1.fn.fn.net.3.bias", "ff2.fn.norm.weight", "ff2.fn.norm.bias",
            "ff2.fn.fn.net.0.weight", "ff2.fn.fn.net.0.bias", "ff2.fn.fn.net.3.weight", "ff2.fn.fn.net.3.bias",
            "attn.norm.weight", "attn.norm.bias", "attn.fn.to_q.weight", "attn.fn.to_kv.weight",
            "attn.fn.to_out.weight", "attn.fn.to_out.bias", "attn.fn.rel_pos_emb.weight", "conv.net.0.weight",
            "conv.net.0.bias", "conv.net.2.weight", "conv.net.2.bias", "conv.net.4.conv.weight",
            "conv.net.4.conv.bias", "conv.net.5.weight", "conv.net.5.bias", "conv.net.5.running_mean",
            "conv.net.5.running_var", "conv.net.7.weight", "conv.net.7.bias", "post_norm.weight", "post_norm.bias"})
        if (need(h, B + k)) return 1;
      if (fold_bn(h, B + "conv.net.5.", h->C2, h->blk[b].bn_scale, h->blk[b].bn_shift, s)) return 1;
      const FT& dw = h->f[B + "conv.net.4.conv.weight"];
      if (dw.n != (size_t)h->C2 * h->ck)
        return fail("afx_finalize: depthwise kernel has %zu elements, expected %d x %d", dw.n, h->C2, h->ck);
      const FT& rp = h->f[B + "attn.fn.rel_pos_emb.weight"];
      if (rp.n != (size_t)1025 * h->dh) return fail("afx_finalize: rel_pos_emb must be (1025, %d)", h->dh);
      KOK(launch_pack_linear(rp.p, 1025, h->dh, 64, h->blk[b].rel_h, h->dt, s));
      if (h->E == 144) {  // parameter blocks of the fused chains (layout: ChainParamOffsets in afx_kernels.h)
        auto put = [&](int st, int off, const std::string& name, int n) -> int {
          HIP_OK(hipMemcpyAsync(h->blk[b].chain_prm[st] + off, h->F(B + name), (size_t)n * 4, hipMemcpyDeviceToDevice, s));
          return 0;
        };
        for (int st = 0; st < 3; ++st) HIP_OK(hipMemsetAsync(h->blk[b].chain_prm[st], 0, (kChainParamFloats + kChainScaleFloats) * 4, s));
        for (int st = 0; st < 3; st += 2) {  // stages 0 and 2 carry a feed-forward module
          const std::string ff = st == 0 ? "ff1" : "ff2";
          if (put(st, CP_FF_G, ff + ".fn.norm.weight", 144) || put(st, CP_FF_B, ff + ".fn.norm.bias", 144) ||
              put(st, CP_FF_B1, ff + ".fn.fn.net.0.bias", 576) || put(st, CP_FF_B2, ff + ".fn.fn.net.3.bias", 144))
            return 1;
        }
        if (put(0, CP_LN2_G, "attn.norm.weight", 144) || put(0, CP_LN2_B, "attn.norm.bias", 144) ||
            put(1, CP_LN2_G, "conv.net.0.weight", 144) || put(1, CP_LN2_B, "conv.net.0.bias", 144) ||
            put(1, CP_BA, "attn.fn.to_out.bias", 144) || put(1, CP_BB, "conv.net.2.bias", 576) ||
            put(2, CP_LN2_G, "post_norm.weight", 144) || put(2, CP_LN2_B, "post_norm.bias", 144) ||
            put(2, CP_BA, "conv.net.7.bias", 144))
          return 1;
        if (h->s3) {  // the weight rows' scales, per output column, behind the parameter block (ChainScaleOffsets)
          ConfBlock& K = h->blk[b];
          auto sc = [&](int st, int off, const void* w, int rows, int kp) -> int {
            HIP_OK(hipMemcpyAsync(K.chain_prm[st] + kChainParamFloats + off, h->wscale(w, rows, kp), (size_t)rows * 4, hipMemcpyDeviceToDevice, s));
            return 0;
          };
          if (sc(0, CS_FF1, K.ff1_w1, h->FF, h->Ep) || sc(0, CS_FF2, K.ff1_w2, h->E, h->FFp) || sc(0, CS_A, K.wqkv, 3 * h->inner, h->Ep) ||
              sc(1, CS_A, K.wout, h->E, h->Ep) || sc(1, CS_B, K.pw1, 2 * h->C2, h->Ep) ||
              sc(2, CS_A, K.pw2, h->E, h->C2p) || sc(2, CS_FF1, K.ff2_w1, h->FF, h->Ep) || sc(2, CS_FF2, K.ff2_w2, h->E, h->FFp))
            return 1;
        }
      }
    }
    if (head_only) {
      if (int rc = s3_layernorm_scales(h, s)) return rc;
      h->finalized = true;
      return 0;
    }
    // BatchNorm2d(1): four scalars -> host (one-off synchronisation)
    float w, b, m, v;
    HIP_OK(hipStreamSynchronize(s));
    HIP_OK(hipMemcpy(&w, h->F("first_bn.weight"), 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&b, h->F("first_bn.bias"), 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&m, h->F("first_bn.running_mean"), 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&v, h->F("first_bn.running_var"), 4, hipMemcpyDeviceToHost));
    h->conf_bn_scale = w / sqrtf(v + kBnEps);
    h->conf_bn_shift = b - m * h->conf_bn_scale;
  } else if (h->cfg.arch == AFX_ARCH_XLSR_AASIST) {
    h->aw.split = h->s3 || h->dt != AFX_DT_FP32;  // exact mode keeps the true-fp32 matrix instruction
    if (const char* m = aasist_finalize(
            h->aw, [&](const std::string& k) { return h->F(k); }, [&](size_t bytes) { return h->dalloc(bytes); }, s))
      return fail("afx_finalize: %s", m);
  }
  if (int rc = s3_layernorm_scales(h, s)) return rc;
  h->finalized = true;
  return 0;
}

// ---------------------------------------------------------------------------------
// workspace carving (identical walk for sizing and for the real call)
// ---------------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t off = 0, largest = 0;
  explicit Carver(void* b) : base((char*)b) {}
  void* take(size_t bytes) {
    if (bytes > largest) largest = bytes;
    off = (off + 255) & ~(size_t)255;
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

struct Ws {
  int T[7];
  void *bufA = nullptr, *bufB = nullptr, *feats_h = nullptr, *xpad = nullptr, *hbuf = nullptr, *qkv = nullptr, *att = nullptr, *ff = nullptr,
       *ssl_h = nullptr;
  float *tmp32 = nullptr, *x = nullptr, *ssl_f = nullptr;
  void* s3planes = nullptr;   // split precision: scratch for the pair form of the current product's A operand
  size_t s3bytes = 0;
  float* gn_stats = nullptr;  // group-norm extractor mode: partial sums + mean / rstd of conv layer 0
  // ragged batch (afx_forward_ragged): valid SSL frames per utterance on the device (null = uniform batch), and the
  // AASIST bucket buffers (utterances of equal length gathered into a uniform sub-batch for the graph back-end)
  int* lens = nullptr;
  float *bucket_f = nullptr, *bucket_logits = nullptr;
  // Conformer
  float *ll32 = nullptr, *xc = nullptr, *qkv32 = nullptr, *glu32 = nullptr;
  void *hc = nullptr, *hid = nullptr, *ao = nullptr, *u = nullptr;
  // AASIST
  AasistWs aa;
};

// Three entry shapes share one walk: the whole path (L > 0 samples), the path from the output of conv layer 5
// (T5 > 0 frames: afx_tail_forward, the streaming mode's per-hop call), the head alone (Tfeat SSL frames).
static size_t carve(const afx_engine* e, int B, int L, int Tfeat, void* base, Ws* w, int T5 = 0, bool ragged = false) {
  Carver c(base);
  const size_t hs = dtype_size(e->dt);
  int T = Tfeat;
  w->lens = nullptr;
  w->bucket_f = w->bucket_logits = nullptr;
  if (L > 0 || T5 > 0) {
    if (L > 0) {
      conv_lengths(L, w->T);
      const size_t n0 = (size_t)B * w->T[0] * kC, n1 = (size_t)B * w->T[1] * kC;
      w->bufA = c.take(n0 * hs);
      w->bufB = c.take(n1 * hs);
      w->tmp32 = (float*)c.take(n1 * 4);
      if (e->cfg.extractor_mode == AFX_EXTRACTOR_GROUP_NORM) w->gn_stats = (float*)c.take(conv0_groupnorm_stats_floats(B, w->T[0]) * 4);
    } else {
      for (int i = 0; i < 5; ++i) w->T[i] = 0;
      w->T[5] = T5;
      w->T[6] = T5 >= kConvK[6] ? (T5 - kConvK[6]) / kConvS[6] + 1 : 0;
      w->bufA = w->bufB = nullptr;
      w->tmp32 = (float*)c.take((size_t)B * (w->T[6] > 0 ? w->T[6] : 1) * kC * 4);
    }
    T = w->T[6];
    w->feats_h = c.take((size_t)B * T * kC * hs);
    w->x = (float*)c.take((size_t)B * T * kD * 4);
    w->xpad = c.take((size_t)B * (T + kPosK) * kD * hs);
    w->hbuf = c.take((size_t)B * T * kD * hs);
    w->qkv = c.take((size_t)B * T * 3 * kD * hs);
    w->att = c.take((size_t)B * T * kD * hs);
    w->ff = c.take((size_t)B * T * kF * hs);
  }
  w->ssl_f = (float*)c.take((size_t)B * T * kD * 4);
  w->ssl_h = c.take((size_t)B * T * kD * hs);
  if (ragged) {
    w->lens = (int*)c.take((size_t)B * 4);
    if (e->cfg.arch == AFX_ARCH_XLSR_AASIST) {
      w->bucket_f = (float*)c.take((size_t)B * T * kD * 4);
      w->bucket_logits = (float*)c.take((size_t)B * 2 * 4);
    }
  }
  if (e->cfg.arch == AFX_ARCH_CONFORMER || e->cfg.arch == AFX_ARCH_CONFORMER_HEAD) {
    const size_t M = (size_t)B * (T + 1);
    w->ll32 = (float*)c.take((size_t)B * T * e->E * 4);
    w->xc = (float*)c.take(M * e->E * 4);
    w->qkv32 = (float*)c.take(M * 3 * e->inner * 4);
    w->glu32 = (float*)c.take(M * 2 * e->C2 * 4);
    w->hc = c.take(M * e->Ep * hs);
    w->hid = c.take(M * e->FFp * hs);
    w->ao = c.take(M * e->Ep * hs);
    w->u = c.take(M * e->C2p * hs);
  } else if (e->cfg.arch == AFX_ARCH_XLSR_AASIST) {
    aasist_carve(B, T, [&](size_t bytes) { return c.take(bytes); }, &w->aa);
  }
  w->s3planes = nullptr;
  w->s3bytes = 0;
  if (e->s3) {  // two fp16 planes of the largest operand = the bytes of its fp32 form (+ the plane alignment)
    size_t big = c.largest;
    if (T5 > 0 && (size_t)B * T5 * kC * 4 > big) big = (size_t)B * T5 * kC * 4;  // (tail mode: the caller's layer-5 frames)
    w->s3bytes = big + 4096;
    w->s3planes = c.take(w->s3bytes);
  }
  return c.off + 256;
}

extern "C" int afx_num_frames(int n) {
  int T[7];
  conv_lengths(n, T);
  return T[6];
}

extern "C" size_t afx_workspace_bytes(afx_handle h, int B, int L) {
  if (!h || B <= 0 || L <= 0) return 0;
  Ws w;
  return carve(h, B, L, 0, nullptr, &w);
}

// ---------------------------------------------------------------------------------
// taps (debug): engine-owned fp32 copies of intermediates
// ---------------------------------------------------------------------------------
__global__ void half_to_f32_kernel(const uint16_t* in, float* out, size_t n, int is_bf16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (is_bf16) {
      out[i] = __uint_as_float((unsigned)in[i] << 16);
    } else {
      _Float16 hv;
      memcpy(&hv, &in[i], 2);
      out[i] = (float)hv;
    }
  }
}
static int tap(afx_engine* e, const char* name, const void* src, size_t n, bool is_half, hipStream_t s) {
  if (!e->taps_on) return 0;
  TapRec& t = e->taps[name];
  if (t.cap < n) {
    if (t.p) (void)hipFree(t.p);
    HIP_OK(hipMalloc((void**)&t.p, n * 4));
    t.cap = n;
  }
  t.n = n;
  if (is_half && e->dt != AFX_DT_FP32) {
    hipLaunchKernelGGL(half_to_f32_kernel, dim3(1024), dim3(256), 0, s, (const uint16_t*)src, t.p, n,
                       e->dt == AFX_DT_BF16 ? 1 : 0);
    HIP_OK(hipGetLastError());
  } else {
    HIP_OK(hipMemcpyAsync(t.p, src, n * 4, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}
extern "C" int afx_tap(afx_handle h, const char* name, float* out, size_t cap, size_t* n_out, void* stream) {
  if (!h || !name) return fail("afx_tap: null argument");
  auto it = h->taps.find(name);
  if (it == h->taps.end()) return fail("afx_tap: no tap named '%s' (enable taps and run a forward first)", name);
  if (n_out) *n_out = it->second.n;
  if (out) {
    if (cap < it->second.n) return fail("afx_tap: buffer too small (%zu < %zu)", cap, it->second.n);
    HIP_OK(hipMemcpyAsync(out, it->second.p, it->second.n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// per-kernel-class timing (bench.py's roofline leg): when profiling is on, every
// launch of the forward is bracketed by hipEvents on the launch stream and summed per
// class afterwards.  Off by default: the normal forward records nothing.
// ---------------------------------------------------------------------------------
enum ProfClass { PC_GEMM128 = 0, PC_GEMM64, PC_GEMM64_DEEP, PC_GEMM256, PC_GEMM_ROWLN, PC_GEMM8_256, PC_GEMM8_ROWLN, PC_GEMM_F32, PC_CONV0, PC_POSCONV, PC_ROWNORM, PC_MHSA, PC_CONF_ATTN, PC_CONF_DWCONV, PC_CONF_CHAIN,
                 PC_MISC, PC_AASIST, PC_COUNT };
static const char* kProfNames[PC_COUNT] = {"gemm_kernel<128x128>", "gemm_kernel<128x64>", "gemm_deep_kernel<128x64>", "gemm_kernel<256x256>",
                                           "gemm_kernel<128x512,rowLN>", "gemm8_kernel<256x256>", "gemm8_kernel<128x512,rowLN>", "gemm_f32_kernel<128x128>", "conv0_kernel", "posconv_kernel", "rownorm_kernel", "mhsa_kernel", "conf_attn_kernel",
                                           "conf_dwconv_kernel", "conf_chain_kernel", "misc", "aasist_head"};
struct ProfRec { int cls; hipEvent_t a, b; double flops; };
struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t ev() {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
};
static thread_local Profiler* t_prof = nullptr;  // the profiler of the engine whose forward runs on this thread
// split precision: the plane scratch of the workspace the forward on this thread runs in (null: another mode)
static thread_local void* t_s3planes = nullptr;
static thread_local size_t t_s3bytes = 0;
// Operand buffers whose ONLY readers are dense products may receive their producer's result directly as the product's A
// operand (the PAIR FORM of the row -- afx_kernels.h -- in place of its fp32 values: same bytes, row by row) -- the LayerNorms,
// the GELU / LayerNorm epilogues of the products and the attention write the operand themselves and the separate split launch
// disappears.  t_s3ok: the buffers of this forward that qualify; t_s3reg: which of them currently hold pair-form rows.
constexpr int kS3Bufs = 10;
struct S3Reg { const void* p; float scale; };  // scale: the power of two the rows were multiplied by before the split (0: fp32 values)
static thread_local const void* t_s3ok[kS3Bufs] = {nullptr};
static thread_local S3Reg t_s3reg[kS3Bufs] = {};
static void s3_begin(std::initializer_list<const void*> ok) {
  int i = 0;
  for (const void* p : ok)
    if (p && i < kS3Bufs) t_s3ok[i++] = p;
  for (; i < kS3Bufs; ++i) t_s3ok[i] = nullptr;
  for (S3Reg& r : t_s3reg) r = S3Reg{nullptr, 0.f};
}
static bool s3_ok(const void* p) {
  if (!p || !t_s3planes) return false;
  for (const void* q : t_s3ok) if (q == p) return true;
  return false;
}
static void s3_set(const void* p, float scale) {  // 0: the buffer holds fp32 values again
  for (S3Reg& r : t_s3reg) if (r.p == p) { r.scale = scale; if (scale == 0.f) r.p = nullptr; return; }
  if (scale == 0.f) return;
  for (S3Reg& r : t_s3reg) if (!r.p) { r = S3Reg{p, scale}; return; }
}
static float s3_pairs_in(const void* p) {  // the scale of the pair-form rows `p` holds, 0 when it holds fp32 values
  for (const S3Reg& r : t_s3reg) if (r.p && r.p == p) return r.scale;
  return 0.f;
}
// per-call context of the forward running on this thread: the engine's profiler, its A/B switches, and (split precision) the
// pair-form scratch of the workspace plus the buffers whose producers may write pair-form rows in place
static thread_local int t_no_deep = 0;
static thread_local const std::unordered_map<const float*, float>* t_ln_scale = nullptr;
static void begin_call(afx_engine* e, const Ws* w);
static void prof_forget(afx_engine* e) {
  if (!e->prof) return;
  for (hipEvent_t ev : e->prof->pool) (void)hipEventDestroy(ev);
  delete e->prof;
  e->prof = nullptr;
}

template <class F>
static const char* timed(int cls, double flops, hipStream_t s, F&& f) {
  Profiler* p = t_prof;
  if (!p || !p->on) return f();
  hipEvent_t a = p->ev(), b = p->ev();
  if (!a || !b) return "profiler: hipEventCreate failed";
  (void)hipEventRecord(a, s);
  const char* m = f();
  (void)hipEventRecord(b, s);
  p->recs.push_back({cls, a, b, flops});
  return m;
}
// Split precision (P_gemm below): x.w ~ xh.wl + xl.wh + xh.wh on the fp16 matrix pipe.  Both operands go in PAIR FORM
// (afx_kernels.h): the weight is stored that way with per-row scales behind it (pack_linear), the fp32 A operand is either
// already there (its producer wrote pair-form rows in place: s3_pairs_in) or the span the product addresses is converted into
// the workspace's scratch; ONE launch then walks the 2 K halfs of a row like a plain fp16 operand and issues three matrix
// instructions per K-step on the fragments it reads anyway (round 3 walked K three times over separate hi / lo planes: 1.5x
// the K-tiles, operand bytes and fragment reads for the same matrix-pipe work).
static const char* P_gemm(const GemmArgs& g_in, int dt, int groups, hipStream_t s) {
  GemmArgs g = g_in;
  g.no_deep = t_no_deep;  // (the engine's "gemm_small_deep" switch)
  const double fl = 2.0 * g.M * g.N * (g.k_algo ? g.k_algo : g.K) * groups;
  // profiler class of a tile id (afx_gemm.hip::gemm_tile_of): 92 = the deep form of the 128x64 tile, a class of its own
  auto cls_of = [](int tile) -> int {
    static const int cls[9] = {PC_GEMM128, PC_GEMM64, PC_GEMM256, PC_GEMM_ROWLN, PC_GEMM256, PC_GEMM256, PC_GEMM_ROWLN, PC_GEMM8_256, PC_GEMM8_ROWLN};
    return tile == 92 ? PC_GEMM64_DEEP : (tile >= 0 && tile < 9 ? cls[tile] : PC_GEMM128);
  };
  if (dt == DT_FP32 && t_s3planes) {
    const bool groups32 = !(g.K % 32 || g.kchunk % 32 || g.a_row % 32 || g.a_batch % 32 || g.g_a % 32 || g.kchunk_stride % 32 || g.ldw % 32 || g.g_w % 32);
    if (!groups32 || ((size_t)g.A & 15)) {  // (rows the pair form cannot address in whole 32-element groups: the fp32 instruction)
      if (g.out_h) s3_set(g.out_h, 0.f);
      return launch_gemm(g, DT_FP32, groups, s);
    }
    GemmArgs q = g;
    float a_scale = s3_pairs_in(g.A);
    if (a_scale == 0.f) {  // convert the span of A this product addresses (fp32 elements from g.A on) into the scratch
      const long last = g.M - 1;
      long span = (last / g.rpb) * g.a_batch + (last % g.rpb) * g.a_row + (long)(g.K / g.kchunk - 1) * g.kchunk_stride + g.kchunk +
                  (long)(groups - 1) * g.g_a;
      span = (span + 31) & ~31L;
      if ((size_t)span * 4 > t_s3bytes) return "split-precision product: the A operand exceeds the pair-form scratch";
      a_scale = kS3ScaleFree;
      if (const char* m = timed(PC_MISC, 0, s, [&] { return launch_split_pairs((const float*)g.A, span, t_s3planes, a_scale, s); })) return m;
      q.A = t_s3planes;
    }
    // every K-side length counts halfs of the pair form: twice the fp32-element value
    q.k1 = g.K;
    q.K = 2 * g.K;
    if (!q.k_algo) q.k_algo = g.K;
    q.kchunk = 2 * g.kchunk;
    q.kchunk_stride = 2 * g.kchunk_stride;
    q.a_row = 2 * g.a_row;
    q.a_batch = 2 * g.a_batch;
    q.g_a = 2 * g.g_a;
    q.ldw = 2 * g.ldw;
    q.g_w = 2 * g.g_w;
    q.pre_scale = (const float*)((const char*)g.W + (size_t)groups * g.N * g.ldw * 4);  // afx_engine::wscale
    q.a_inv = 1.0f / a_scale;
    // the result as the NEXT product's A operand, where only products read the buffer (pair-form rows in place of fp32 rows);
    // a LayerNorm epilogue bounds its output (the larger scale), anything else is unbounded in a trained checkpoint
    if (g.out_h) {
      const bool pairs_out = s3_ok(g.out_h) && (g.N & 7) == 0 && (g.g_n & 7) == 0 && (g.ldo_h & 31) == 0 && g.out_h != g.A;
      q.oh_pairs = pairs_out ? 1 : 0;
      q.oh_scale = g.ln_gamma ? kS3ScaleBounded : kS3ScaleFree;
      s3_set(g.out_h, pairs_out ? q.oh_scale : 0.f);
    }
    const int tile = gemm_tile_of(q, groups);
    return timed(cls_of(tile), fl, s, [&] { return launch_gemm(q, DT_FP16X3, groups, s); });
  }
  return timed(dt == DT_FP32 ? PC_GEMM_F32 : cls_of(gemm_tile_of(g, groups)), fl, s, [&] { return launch_gemm(g, dt, groups, s); });
}
static const char* P_rownorm(const RowNormArgs& a_in, int dt, hipStream_t s) {
  RowNormArgs a = a_in;
  if (a.out_h && t_s3planes) {  // split precision: the LayerNorm writes the next product's A operand (pair-form rows) itself
    const bool pairs_out = dt == DT_FP32 && s3_ok(a.out_h) && (a.ldo_h & 31) == 0 && a.out_h != (const void*)a.x;
    a.oh_pairs = pairs_out ? 1 : 0;
    a.oh_scale = kS3ScaleBounded;  // |LayerNorm output| <= sqrt(C) max|gamma| + max|beta|: the scale finalize chose for this LayerNorm
    if (t_ln_scale) {
      auto it = t_ln_scale->find(a.gamma);
      if (it != t_ln_scale->end()) a.oh_scale = it->second;
    }
    s3_set(a.out_h, pairs_out ? a.oh_scale : 0.f);
  }
  return timed(PC_ROWNORM, 0, s, [&] { return launch_rownorm(a, dt, s); });
}

static void begin_call(afx_engine* e, const Ws* w) {
  t_prof = e->prof;
  t_no_deep = e->gemm_small_deep ? 0 : 1;
  t_ln_scale = &e->ln_scale;
  t_s3planes = w ? w->s3planes : nullptr;
  t_s3bytes = w ? w->s3bytes : 0;
  if (w) s3_begin({w->bufA, w->bufB, w->feats_h, w->hbuf, w->att, w->ff, w->xpad, w->hc, w->ssl_h, w->hid});
  else s3_begin({});
}

extern "C" int afx_profile_begin(afx_handle h) {
  if (!h) return fail("afx_profile_begin: null handle");
  if (!h->prof) h->prof = new Profiler();
  Profiler& p = *h->prof;
  p.on = true;
  p.recs.clear();
  p.used = 0;
  return 0;
}
extern "C" int afx_profile_end(afx_handle h, int n, double* ms, double* flops, long long* launches) {
  if (!h || n < PC_COUNT || !ms || !flops || !launches) return fail("afx_profile_end: need %d slots", (int)PC_COUNT);
  if (!h->prof) return fail("afx_profile_end: afx_profile_begin was not called on this handle");
  Profiler& p = *h->prof;
  for (int i = 0; i < n; ++i) { ms[i] = 0; flops[i] = 0; launches[i] = 0; }
  for (const ProfRec& r : p.recs) {
    HIP_OK(hipEventSynchronize(r.b));
    float t = 0;
    HIP_OK(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.cls] += t;
    flops[r.cls] += r.flops;
    launches[r.cls] += 1;
  }
  p.on = false;
  p.recs.clear();
  p.used = 0;
  return 0;
}
extern "C" int afx_profile_num_classes(void) { return PC_COUNT; }
extern "C" const char* afx_profile_class_name(int c) { return c >= 0 && c < PC_COUNT ? kProfNames[c] : ""; }

// ---------------------------------------------------------------------------------
// forward pieces
// ---------------------------------------------------------------------------------
static GemmArgs plain_gemm(const void* A, long lda, const void* W, long ldw, int M, int N, int K) {
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.A = A; g.W = W; g.M = M; g.N = N; g.K = K;
  g.rpb = M; g.a_batch = 0; g.a_row = lda;
  g.kchunk = K; g.kchunk_stride = 0; g.ldw = ldw;
  g.alpha = 1.f; g.act = ACT_NONE;
  g.o_batch_rows = M; g.oh_batch_rows = M;
  return g;
}

static RowNormArgs plain_norm(const float* x, long ldx, int rows, int C, const float* g, const float* b) {
  RowNormArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.ldx = ldx; a.rows = rows; a.C = C; a.gamma = g; a.beta = b; a.eps = kLnEps; a.act = ACT_NONE;
  a.rpb = rows; a.o_batch_rows = rows;
  return a;
}

#define launch_gemm P_gemm
#define launch_rownorm P_rownorm

// l5: null = start from the waveform; else the output of conv layer 5, (B, T[5], 512) operand type (tail mode)
// l5_batch: elements between two utterances of l5 (0 = packed, T[5] * 512): a streaming caller keeps its window inside a longer ring
static int run_trunk(afx_engine* e, const float* wave, int B, int L, Ws& w, hipStream_t s, const void* l5 = nullptr, long l5_batch = 0) {
  const int dt = e->dt;
  const int* T = w.T;
  if (T[6] < 1) return fail("afx_forward: %d samples are too few for one output frame (need >= 400)", L);
  auto cf = [&](int i, const char* leaf) {
    return e->F("ssl.feature_extractor.conv_layers." + std::to_string(i) + leaf);
  };
  // layer 0: waveform -> (B,T0,512) operand type, normalisation + GELU fused
  const bool gn = e->cfg.extractor_mode == AFX_EXTRACTOR_GROUP_NORM;
  if (!l5)
    KOK(timed(PC_CONV0, 2.0 * B * T[0] * kC * kConvK[0], s, [&] {
      if (gn)  // wav2vec2-base: GroupNorm over time per (utterance, channel), two passes over the cheap convolution
        return launch_conv0_groupnorm(wave, B, L, T[0], cf(0, ".0.weight"), cf(0, ".2.weight"), cf(0, ".2.bias"), kLnEps,
                                      w.gn_stats, w.bufA, dt, s);
      if (e->s3) {  // split precision: the same matrix-core kernel as the fp16 engines; its rows leave as conv layer 1's pair-form operand
        float sc = 0.f;
        if (s3_ok(w.bufA)) {
          sc = kS3ScaleBounded;  // (LayerNorm + GELU output: bounded by the LayerNorm's gains -- the scale finalize chose for it)
          auto it = e->ln_scale.find(cf(0, ".2.1.weight"));
          if (it != e->ln_scale.end()) sc = it->second;
        }
        s3_set(w.bufA, sc);
        return launch_conv0(wave, B, L, T[0], cf(0, ".0.weight"), cf(0, ".0.bias"), cf(0, ".2.1.weight"), cf(0, ".2.1.bias"),
                            e->cfg.pre_emphasis, e->cfg.pre_emphasis_coef, w.bufA, DT_FP16X3, s, e->conv0pack, sc);
      }
      return launch_conv0(wave, B, L, T[0], cf(0, ".0.weight"), cf(0, ".0.bias"), cf(0, ".2.1.weight"),
                          cf(0, ".2.1.bias"), e->cfg.pre_emphasis, e->cfg.pre_emphasis_coef, w.bufA, dt, s, e->conv0pack);
    }));
  // layers 1..6: conv-as-GEMM on a row-complete tile, LayerNorm(512) + GELU fused into the
  // epilogue (the pre-norm fp32 activations never leave the registers)
  void* in = l5 ? const_cast<void*>(l5) : w.bufA;
  void* out = w.bufB;
  for (int i = l5 ? 6 : 1; i < 7; ++i) {
    const int M = B * T[i], K = kConvK[i] * kC;
    GemmArgs g = plain_gemm(in, 0, e->convw[i], K, M, kC, K);
    g.rpb = T[i]; g.a_batch = (l5 && i == 6 && l5_batch) ? l5_batch : (long)T[i - 1] * kC; g.a_row = (long)kConvS[i] * kC;
    g.o_batch_rows = T[i]; g.oh_batch_rows = T[i];
    g.bias = gn ? nullptr : cf(i, ".0.bias");
    g.act = ACT_GELU;
    if (gn) {  // wav2vec2-base: conv -> GELU, nothing to normalise in layers 1-6 (plain product, GELU epilogue)
      if (i < 6) {
        g.out_h = out; g.ldo_h = kC;
      } else {
        g.out_f = w.tmp32; g.ldo_f = kC;
      }
      KOK(launch_gemm(g, dt, 1, s));
    } else if (e->fuse_conv_ln && dt != DT_FP32) {
      // (split precision takes the two-kernel form below: measured, the row-complete tile with the pair-form walk and the fp32
      // erf-GELU in its epilogue is SLOWER than the 256-wide tile + a LayerNorm pass that writes the next layer's pair-form
      // operand -- student 12.9 against 12.3 ms per forward, teacher 9.0 / 8.75: profiles/r04_s3_knobs.txt)
      g.ln_gamma = cf(i, ".2.1.weight"); g.ln_beta = cf(i, ".2.1.bias"); g.ln_eps = kLnEps;
      if (i < 6) {
        g.out_h = out; g.ldo_h = kC;
      } else {
        g.out_f = w.tmp32; g.ldo_f = kC;  // fp32: the feature LayerNorm follows
      }
      KOK(launch_gemm(g, dt, 1, s));
    } else {  // two-kernel form (kept for A/B measurements)
      g.act = ACT_NONE;
      g.out_f = w.tmp32; g.ldo_f = kC;
      KOK(launch_gemm(g, dt, 1, s));
      RowNormArgs n = plain_norm(w.tmp32, kC, M, kC, cf(i, ".2.1.weight"), cf(i, ".2.1.bias"));
      n.act = ACT_GELU;
      if (i < 6) {
        n.out_h = out; n.ldo_h = kC;
      } else {
        n.out_f = w.tmp32; n.ldo_f = kC;  // in place: a row is register-resident before it is written
      }
      KOK(launch_rownorm(n, dt, s));
    }
    void* t = in; in = out; out = t;
  }
  const int Tt = T[6], M = B * Tt;
  if (tap(e, "conv", w.tmp32, (size_t)M * kC, false, s)) return 1;
  // feature LayerNorm(512) -> operand type
  {
    RowNormArgs n = plain_norm(w.tmp32, kC, M, kC, e->F("ssl.layer_norm.weight"), e->F("ssl.layer_norm.bias"));
    n.out_h = w.feats_h; n.ldo_h = kC;
    KOK(launch_rownorm(n, dt, s));
  }
  // post_extract_proj: fp32 residual stream x + operand copy into the time-padded buffer
  {
    GemmArgs g = plain_gemm(w.feats_h, kC, e->projw, kC, M, kD, kC);
    g.rpb = Tt; g.a_batch = (long)Tt * kC; g.a_row = kC;
    g.bias = e->F("ssl.post_extract_proj.bias");
    g.out_f = w.x; g.ldo_f = kD; g.o_batch_rows = Tt; g.o_row_off = 0;
    g.out_h = w.xpad; g.ldo_h = kD; g.oh_batch_rows = Tt + kPosK; g.oh_row_off = kPosPad;
    KOK(launch_gemm(g, dt, 1, s));
    // (ragged batch: the frames past an utterance's own length are zeroed too -- alone, its positional conv would
    // see zero padding there)
    KOK(timed(PC_MISC, 0, s, [&] { return launch_zero_pad_rows(w.xpad, B, Tt, kD, kPosPad, kPosK - kPosPad, dt, s, w.lens); }));
  }
  if (tap(e, "proj", w.x, (size_t)M * kD, false, s)) return 1;
  // positional conv (grouped, k=128) + GELU, added to x in place
  if (e->posconv_sliding && dt != DT_FP32 && Tt <= 224) {  // (reads the zero-padded operand copy: ragged batches need nothing more)
    PosConvArgs pc;
    memset(&pc, 0, sizeof pc);
    pc.xpad = w.xpad; pc.xpad_batch = (long)(Tt + kPosK) * kD; pc.W = e->posw; pc.bias = e->F("ssl.encoder.pos_conv.0.bias");
    pc.x = w.x; pc.B = B; pc.T = Tt;
    KOK(timed(PC_POSCONV, 2.0 * M * kD * (kD / kPosG) * kPosK, s, [&] { return launch_posconv(pc, dt, s); }));
  } else {
    const int cpg = kD / kPosG;
    GemmArgs g = plain_gemm(w.xpad, 0, e->posw, (long)cpg * kPosK, M, cpg, cpg * kPosK);
    g.rpb = Tt; g.a_batch = (long)(Tt + kPosK) * kD; g.a_row = kD;
    g.kchunk = cpg; g.kchunk_stride = kD;
    g.g_a = cpg; g.g_w = (long)cpg * cpg * kPosK; g.g_n = cpg;
    g.bias = e->F("ssl.encoder.pos_conv.0.bias");
    g.act = ACT_GELU;
    g.resid = w.x; g.ldr = kD;
    g.out_f = w.x; g.ldo_f = kD; g.o_batch_rows = Tt; g.oh_batch_rows = Tt;
    KOK(launch_gemm(g, dt, kPosG, s));
  }
  if (tap(e, "pos", w.x, (size_t)M * kD, false, s)) return 1;
  // transformer layers (pre-LN)
  auto resid_product = [&](const void* A, long lda, const void* W, int K, const float* bias) -> const char* {
    GemmArgs o = plain_gemm(A, lda, W, K, M, kD, K);
    o.bias = bias; o.resid = w.x; o.ldr = kD; o.out_f = w.x; o.ldo_f = kD;
    return launch_gemm(o, dt, 1, s);
  };
  for (int l = 0; l < e->cfg.n_layers; ++l) {
    const std::string P = "ssl.encoder.layers." + std::to_string(l) + ".";
    RowNormArgs n1 = plain_norm(w.x, kD, M, kD, e->F(P + "self_attn_layer_norm.weight"), e->F(P + "self_attn_layer_norm.bias"));
    n1.out_h = w.hbuf; n1.ldo_h = kD;
    KOK(launch_rownorm(n1, dt, s));
    GemmArgs q = plain_gemm(w.hbuf, kD, e->wqkv[l], kD, M, 3 * kD, kD);
    q.bias = e->bqkv[l];
    q.out_h = w.qkv; q.ldo_h = 3 * kD;
    KOK(launch_gemm(q, dt, 1, s));
    KOK(timed(PC_MHSA, 4.0 * B * kH * (double)Tt * Tt * 64, s, [&] {
      if (e->s3 && Tt <= 224) {  // split precision: the matrix-core form; its output goes out as the output projection's A planes
        const bool pairs = s3_ok(w.att);
        s3_set(w.att, pairs ? kS3ScaleFree : 0.f);
        return launch_mhsa_split((const float*)w.qkv, (float*)w.att, B, Tt, kH, s, w.lens, pairs, kS3ScaleFree);
      }
      if (e->s3) s3_set(w.att, 0.f);
      return launch_mhsa(w.qkv, w.att, B, Tt, kH, dt, s, w.lens);
    }));
    KOK(resid_product(w.att, kD, e->wo[l], kD, e->F(P + "self_attn.out_proj.bias")));
    RowNormArgs n2 = plain_norm(w.x, kD, M, kD, e->F(P + "final_layer_norm.weight"), e->F(P + "final_layer_norm.bias"));
    n2.out_h = w.hbuf; n2.ldo_h = kD;
    KOK(launch_rownorm(n2, dt, s));
    GemmArgs f1 = plain_gemm(w.hbuf, kD, e->w1[l], kD, M, kF, kD);
    f1.bias = e->F(P + "fc1.bias"); f1.act = ACT_GELU;
    f1.out_h = w.ff; f1.ldo_h = kF;
    KOK(launch_gemm(f1, dt, 1, s));
    KOK(resid_product(w.ff, kF, e->w2[l], kF, e->F(P + "fc2.bias")));
    if (e->taps_on) {
      const std::string nm = "layer" + std::to_string(l);
      if (tap(e, nm.c_str(), w.x, (size_t)M * kD, false, s)) return 1;
    }
  }
  // final encoder LayerNorm -> fp32 features (API output / AASIST input) + operand copy (LL GEMM)
  RowNormArgs nf = plain_norm(w.x, kD, M, kD, e->F("ssl.encoder.layer_norm.weight"), e->F("ssl.encoder.layer_norm.bias"));
  nf.out_f = w.ssl_f; nf.ldo_f = kD;
  nf.out_h = w.ssl_h; nf.ldo_h = kD;
  nf.nonfinite = e->nonfinite;  // overflow guard: an operand copy that left fp16's range anywhere in the trunk ends here as inf / NaN
  KOK(launch_rownorm(nf, dt, s));
  if (tap(e, "ssl", w.ssl_f, (size_t)M * kD, false, s)) return 1;
  return 0;
}

// Conformer head from SSL features already in w.ssl_h (operand type)
// tokens: null = from the SSL features in w.ssl_h (Model / MyModel.forward); else the (B,T,E) fp32 rows MyConformer.forward
// is given (models/conformer_baseline.py:22-29).  embedding (device (B,E) fp32, may be null) receives token 0 of every utterance.
static int run_conformer(afx_engine* e, int B, int T, Ws& w, float* logits, hipStream_t s, const float* tokens = nullptr,
                         float* embedding = nullptr) {
  const int dt = e->dt, E = e->E, Ep = e->Ep, N = T + 1, M = B * N;
  if (tokens) {  // class token + the rows as they are
    KOK(timed(PC_MISC, 0, s, [&] { return launch_conf_tokens(tokens, e->F("conformer.class_token"), 1.f, 0.f, B, T, E, w.xc, s, true); }));
  } else {
    // LL -> BatchNorm2d(1) -> SELU -> class token
    GemmArgs ll = plain_gemm(w.ssl_h, kD, e->conf_ll, kD, B * T, E, kD);
    ll.bias = e->F("LL.bias");
    ll.out_f = w.ll32; ll.ldo_f = E;
    KOK(launch_gemm(ll, dt, 1, s));
    KOK(timed(PC_MISC, 0, s, [&] {
      return launch_conf_tokens(w.ll32, e->F("conformer.class_token"), e->conf_bn_scale, e->conf_bn_shift, B, T, E, w.xc, s);
    }));
  }
  if (tap(e, "tokens", w.xc, (size_t)M * E, false, s)) return 1;
  // K-padding columns of the operand buffers must read as zero (the fused chains only read past
  // the real columns of the attention output: 144 -> 160)
  const size_t hs = dtype_size(dt);
  // (split precision: the chains take fp32 operand rows and the pair-form weights; dt is DT_FP32 for every other kernel)
  const bool fused = e->fuse_conformer && (dt != DT_FP32 || e->s3) && E == 144 && e->inner == E && e->FFp == 4 * E && e->C2 == 2 * E;
  const int chain_dt = e->s3 ? DT_FP16X3 : dt;
  HIP_OK(hipMemsetAsync(w.ao, 0, (size_t)M * Ep * hs, s));
  if (!fused) {
    HIP_OK(hipMemsetAsync(w.hc, 0, (size_t)M * Ep * hs, s));
    HIP_OK(hipMemsetAsync(w.u, 0, (size_t)M * e->C2p * hs, s));
    if (e->FFp != e->FF) HIP_OK(hipMemsetAsync(w.hid, 0, (size_t)M * e->FFp * hs, s));
  }
  for (int b = 0; fused && b < e->nblk; ++b) {
    // three register-resident row chains around the attention and the depthwise conv
    const std::string P = "conformer.encoder_blocks." + std::to_string(b) + ".";
    ConfBlock& K = e->blk[b];
    ConfChainArgs c;
    memset(&c, 0, sizeof c);
    c.M = M; c.E = E; c.Ep = Ep; c.FFp = e->FFp;
    c.x_in = w.xc; c.x_out = w.xc;  // in place: a wave reads and writes only its own 16 rows
    auto ff = [&](void* w1, void* w2) { c.ff_w1 = w1; c.ff_w2 = w2; };
    const double ff_fl = 2.0 * M * E * e->FF * 2;
    ff(K.ff1_w1, K.ff1_w2);
    c.params = K.chain_prm[0];
    c.w_a = K.wqkv; c.ld_w_a = Ep;
    c.out2 = w.qkv32; c.ld_out2 = 3 * e->inner;
    KOK(timed(PC_CONF_CHAIN, ff_fl + 2.0 * M * E * 3 * e->inner, s, [&] { return launch_conf_chain(c, 0, chain_dt, s); }));
    KOK(timed(PC_CONF_ATTN, 6.0 * B * e->heads * (double)N * N * e->dh, s, [&] {
      if (e->conf_attn_mfma && dt != DT_FP32 && e->dh == 36)
        return launch_conf_attn_mfma(w.qkv32, 3 * e->inner, w.qkv32 + e->inner, 3 * e->inner, K.rel_h, 512, B, N, e->heads,
                                     e->dh, w.ao, Ep, dt, s, w.lens, 1);
      if (e->conf_attn_mfma && e->s3 && e->dh == 36 && N <= 209)  // split precision: the one-pass kernel on hi / lo halves, fp32 rows out
        return launch_conf_attn_split(w.qkv32, 3 * e->inner, w.qkv32 + e->inner, 3 * e->inner, (const float*)K.rel_h, 512, B, N,
                                      e->heads, e->dh, (float*)w.ao, Ep, s, w.lens, 1);
      return launch_conf_attn(w.qkv32, 3 * e->inner, w.qkv32 + e->inner, 3 * e->inner,
                              e->F(P + "attn.fn.rel_pos_emb.weight"), 512, B, N, e->heads, e->dh, w.ao, Ep, dt, s, w.lens, 1);
    }));
    c.in_h = w.ao; c.ld_in_h = Ep;
    c.params = K.chain_prm[1];
    c.w_a = K.wout;
    c.w_b = K.pw1;
    c.out2 = w.glu32; c.ld_out2 = 2 * e->C2;
    KOK(timed(PC_CONF_CHAIN, 2.0 * M * E * (e->inner + 2 * e->C2), s, [&] { return launch_conf_chain(c, 1, chain_dt, s); }));
    KOK(timed(PC_CONF_DWCONV, 2.0 * B * N * e->C2 * e->ck, s, [&] {
      return launch_conf_dwconv(w.glu32, 2 * e->C2, e->F(P + "conv.net.4.conv.weight"),
                                e->F(P + "conv.net.4.conv.bias"), K.bn_scale, K.bn_shift, B, N, e->C2, e->ck, w.u,
                                e->C2p, dt, s, w.lens, 1);
    }));
    c.in_h = w.u; c.ld_in_h = e->C2p;
    c.params = K.chain_prm[2];
    c.w_a = K.pw2; c.ld_w_a = e->C2p;
    ff(K.ff2_w1, K.ff2_w2);
    c.w_b = nullptr; c.out2 = nullptr;
    KOK(timed(PC_CONF_CHAIN, ff_fl + 2.0 * M * e->C2 * E, s, [&] { return launch_conf_chain(c, 2, chain_dt, s); }));
    if (e->taps_on) {
      const std::string nm = "block" + std::to_string(b);
      if (tap(e, nm.c_str(), w.xc, (size_t)M * E, false, s)) return 1;
    }
  }
  for (int b = 0; !fused && b < e->nblk; ++b) {
    const std::string P = "conformer.encoder_blocks." + std::to_string(b) + ".";
    ConfBlock& K = e->blk[b];
    auto norm_to_h = [&](const char* nm) -> const char* {
      RowNormArgs n = plain_norm(w.xc, E, M, E, e->F(P + nm + ".weight"), e->F(P + nm + ".bias"));
      n.out_h = w.hc; n.ldo_h = Ep;
      return launch_rownorm(n, dt, s);
    };
    auto feed_forward = [&](const char* ff, void* w1, void* w2) -> const char* {
      if (const char* m = norm_to_h((std::string(ff) + ".fn.norm").c_str())) return m;
      GemmArgs a = plain_gemm(w.hc, Ep, w1, Ep, M, e->FF, Ep);
      a.k_algo = E;
      a.bias = e->F(P + ff + ".fn.fn.net.0.bias"); a.act = ACT_SWISH;
      a.out_h = w.hid; a.ldo_h = e->FFp;
      if (const char* m = launch_gemm(a, dt, 1, s)) return m;
      GemmArgs c = plain_gemm(w.hid, e->FFp, w2, e->FFp, M, E, e->FFp);
      c.k_algo = e->FF;
      c.bias = e->F(P + ff + ".fn.fn.net.3.bias"); c.alpha = 0.5f;
      c.resid = w.xc; c.ldr = E; c.out_f = w.xc; c.ldo_f = E;
      return launch_gemm(c, dt, 1, s);
    };
    KOK(feed_forward("ff1", K.ff1_w1, K.ff1_w2));
    // attention
    KOK(norm_to_h("attn.norm"));
    GemmArgs q = plain_gemm(w.hc, Ep, K.wqkv, Ep, M, 3 * e->inner, Ep);
    q.k_algo = E;
    q.out_f = w.qkv32; q.ldo_f = 3 * e->inner;
    KOK(launch_gemm(q, dt, 1, s));
    KOK(timed(PC_CONF_ATTN, 6.0 * B * e->heads * (double)N * N * e->dh, s, [&] {
      if (e->conf_attn_mfma && dt != DT_FP32 && e->dh == 36)
        return launch_conf_attn_mfma(w.qkv32, 3 * e->inner, w.qkv32 + e->inner, 3 * e->inner, K.rel_h, 512, B, N, e->heads,
                                     e->dh, w.ao, Ep, dt, s, w.lens, 1);
      if (e->conf_attn_mfma && e->s3 && e->dh == 36 && N <= 209)  // split precision: the one-pass kernel on hi / lo halves, fp32 rows out
        return launch_conf_attn_split(w.qkv32, 3 * e->inner, w.qkv32 + e->inner, 3 * e->inner, (const float*)K.rel_h, 512, B, N,
                                      e->heads, e->dh, (float*)w.ao, Ep, s, w.lens, 1);
      return launch_conf_attn(w.qkv32, 3 * e->inner, w.qkv32 + e->inner, 3 * e->inner,
                              e->F(P + "attn.fn.rel_pos_emb.weight"), 512, B, N, e->heads, e->dh, w.ao, Ep, dt, s, w.lens, 1);
    }));
    GemmArgs o = plain_gemm(w.ao, Ep, K.wout, Ep, M, E, Ep);
    o.k_algo = e->inner;
    o.bias = e->F(P + "attn.fn.to_out.bias");
    o.resid = w.xc; o.ldr = E; o.out_f = w.xc; o.ldo_f = E;
    KOK(launch_gemm(o, dt, 1, s));
    // conv module
    KOK(norm_to_h("conv.net.0"));
    GemmArgs p1 = plain_gemm(w.hc, Ep, K.pw1, Ep, M, 2 * e->C2, Ep);
    p1.k_algo = E;
    p1.bias = e->F(P + "conv.net.2.bias");
    p1.out_f = w.glu32; p1.ldo_f = 2 * e->C2;
    KOK(launch_gemm(p1, dt, 1, s));
    KOK(timed(PC_CONF_DWCONV, 2.0 * B * N * e->C2 * e->ck, s, [&] {
      return launch_conf_dwconv(w.glu32, 2 * e->C2, e->F(P + "conv.net.4.conv.weight"),
                                e->F(P + "conv.net.4.conv.bias"), K.bn_scale, K.bn_shift, B, N, e->C2, e->ck, w.u,
                                e->C2p, dt, s, w.lens, 1);
    }));
    GemmArgs p2 = plain_gemm(w.u, e->C2p, K.pw2, e->C2p, M, E, e->C2p);
    p2.k_algo = e->C2;
    p2.bias = e->F(P + "conv.net.7.bias");
    p2.resid = w.xc; p2.ldr = E; p2.out_f = w.xc; p2.ldo_f = E;
    KOK(launch_gemm(p2, dt, 1, s));
    KOK(feed_forward("ff2", K.ff2_w1, K.ff2_w2));
    RowNormArgs pn = plain_norm(w.xc, E, M, E, e->F(P + "post_norm.weight"), e->F(P + "post_norm.bias"));
    pn.out_f = w.xc; pn.ldo_f = E;
    KOK(launch_rownorm(pn, dt, s));
    if (e->taps_on) {
      const std::string nm = "block" + std::to_string(b);
      if (tap(e, nm.c_str(), w.xc, (size_t)M * E, false, s)) return 1;
    }
  }
  KOK(timed(PC_MISC, 0, s, [&] {
    return launch_small_linear(w.xc, (long)N * E, B, E, e->F("conformer.fc5.weight"), e->F("conformer.fc5.bias"), 2,
                               logits, s, e->nonfinite + 1);
  }));
  if (embedding)  // x[:, 0, :] (models/conformer_baseline.py:27)
    HIP_OK(hipMemcpy2DAsync(embedding, (size_t)E * 4, w.xc, (size_t)N * E * 4, (size_t)E * 4, B, hipMemcpyDeviceToDevice, s));
  return 0;
}

static int run_head(afx_engine* e, int B, int T, Ws& w, float* logits, hipStream_t s) {
  if (e->cfg.arch == AFX_ARCH_CONFORMER) return run_conformer(e, B, T, w, logits, s);
  if (e->cfg.arch == AFX_ARCH_XLSR_AASIST) {
    if (const char* m = timed(PC_AASIST, 2.0 * B * 1.399e9 / 2, s, [&] { return aasist_forward(e->aw, w.ssl_f, B, T, w.aa, logits, s, e->nonfinite + 1); }))
      return fail("%s", m);
    if (e->taps_on) {
      if (tap(e, "e_S", w.aa.eS, (size_t)B * 42 * 64, false, s)) return 1;
      if (tap(e, "e_T", w.aa.eT, (size_t)B * (T / 3) * 64, false, s)) return 1;
      if (tap(e, "hidden", w.aa.hidden, (size_t)B * 160, false, s)) return 1;
      for (int i = 0; i < w.aa.dbg_count; ++i)
        if (tap(e, w.aa.dbg_name[i], w.aa.dbg_ptr[i], w.aa.dbg_n[i], false, s)) return 1;
    }
    return 0;
  }
  return fail("afx_forward: this handle is an SSL feature extractor; use afx_ssl_forward");
}

__global__ void f32_to_half_kernel(const float* in, uint16_t* out, size_t n, int is_bf16);  // (defined below)

// ---------------------------------------------------------------------------------
// KV-cached streaming mode (BASELINE config 5 as named: "250 ms chunks with cached SSL-encoder KV state").
// NOT A REFERENCE FUNCTION.  The reference's trunk is bidirectional over the clip (full self-attention, a centred
// k = 128 positional conv, models/fe.py:17-21), so an encoder that sees every frame ONCE, when its chunk arrives, is
// a different model; SURVEY.md section 7 scopes it as a labelled mode whose parity target is this build's own offline
// restatement of the same function (oracle/streaming.py), not the reference.  The function, per chunk c of n_c frames
// (the frames of conv layer 6 that became computable with the chunk's samples):
//   * conv feature extractor, feature LayerNorm, projection: frame-local -- exact;
//   * positional conv: frame t of chunk c sees projected frames [t - 64, end of chunk c] (left context cached, the
//     right context beyond the chunk is zero, as beyond a clip's end);
//   * every transformer layer: the chunk's frames are the queries; keys / values are the chunk itself and the cached
//     K / V of the 15 chunks before it (16 chunks = 4 s of context), written once when their chunk passed;
//   * final LayerNorm; the back-end (AASIST / Conformer head) scores the window of the last <= 200 feature frames.
// State per stream: 24 x 256 x 3072 halfs of [q | k | v] rows (16-slot groups, one per chunk), 64 projected frames, the
// feature window.  The whole step runs on new frames only: S x n_c rows through the products, one query tile per
// (stream, head) in the attention.
// ---------------------------------------------------------------------------------
constexpr int kKvGroups = 16, kKvSlots = 256, kKvHist = 64, kKvFeat = 208, kKvWindow = 200;
struct afx_kv {
  afx_engine* e;
  int S;
  long hop = 0;
  int cnt[kKvGroups];
  int nfeat = 0, pp = 0;
  void* rings = nullptr;   // (layers, S, 256, 3072) operand type
  void* hist = nullptr;    // (S, 64, 1024) operand type: the newest projected frames (positional conv's left context)
  float* feat[2] = {nullptr, nullptr};  // (S, 208, 1024) fp32, right-aligned window, ping-pong
};
struct KvWs {
  void *feats_h, *xpad, *hbuf, *att, *ff;
  float *xp, *x;
  void* s3planes = nullptr;  // split precision: pair-form scratch of the chunk's products
  size_t s3bytes = 0;
  Ws head;
};
static size_t kv_carve(const afx_engine* e, int S, int n, int Th, void* base, KvWs* k) {
  Carver c(base);
  const size_t hs = dtype_size(e->dt), M = (size_t)S * n, Tp = kKvHist + n;
  k->feats_h = c.take(M * kC * hs);
  k->xpad = c.take((size_t)S * (Tp + kPosK) * kD * hs);
  k->xp = (float*)c.take((size_t)S * Tp * kD * 4);
  k->x = (float*)c.take(M * kD * 4);
  k->hbuf = c.take(M * kD * hs);
  k->att = c.take((size_t)S * 16 * kD * hs);
  k->ff = c.take(M * kF * hs);
  k->s3planes = nullptr;
  k->s3bytes = 0;
  if (e->s3) {
    k->s3bytes = c.largest + 4096;
    k->s3planes = c.take(k->s3bytes);
  }
  c.off = (c.off + 255) & ~(size_t)255;
  const size_t head_bytes = carve(e, S, 0, Th, base ? (char*)base + c.off : nullptr, &k->head);
  return c.off + head_bytes + 256;
}
extern "C" int afx_kv_create(afx_handle h, int n_streams, afx_kv** out) {
  if (!h || !out || n_streams <= 0) return fail("afx_kv_create: bad argument");
  if (!h->finalized) return fail("afx_kv_create: weights not finalized");
  if (h->cfg.arch != AFX_ARCH_XLSR_AASIST && h->cfg.arch != AFX_ARCH_CONFORMER) return fail("afx_kv_create: a handle with a trunk and a back-end");
  if (h->dt == DT_FP32 && !h->s3) return fail("afx_kv_create: the KV-cached mode runs the half-precision kernels or split precision (fp16 / bf16 / fp16x3 engines)");
  if (h->cfg.extractor_mode != AFX_EXTRACTOR_LAYER_NORM || h->cfg.pre_emphasis) return fail("afx_kv_create: layer_norm extractor without fused pre-emphasis");
  afx_kv* k = new afx_kv();
  k->e = h;
  k->S = n_streams;
  for (int i = 0; i < kKvGroups; ++i) k->cnt[i] = 0;
  const size_t hs = h->hsz, ring = (size_t)h->cfg.n_layers * n_streams * kKvSlots * 3 * kD * hs, hist = (size_t)n_streams * kKvHist * kD * hs,
               feat = (size_t)n_streams * kKvFeat * kD * 4;
  bool ok = hipMalloc(&k->rings, ring) == hipSuccess && hipMalloc(&k->hist, hist) == hipSuccess &&
            hipMalloc((void**)&k->feat[0], feat) == hipSuccess && hipMalloc((void**)&k->feat[1], feat) == hipSuccess;
  // the history starts as silence-before-the-stream (zero left padding); ring slots are masked until written
  ok = ok && hipMemset(k->rings, 0, ring) == hipSuccess && hipMemset(k->hist, 0, hist) == hipSuccess &&
       hipMemset(k->feat[0], 0, feat) == hipSuccess && hipMemset(k->feat[1], 0, feat) == hipSuccess;
  if (!ok) {
    (void)hipFree(k->rings); (void)hipFree(k->hist); (void)hipFree(k->feat[0]); (void)hipFree(k->feat[1]);
    delete k;
    return fail("afx_kv_create: device allocation failed (%zu bytes per stream)", (ring + hist + 2 * feat) / n_streams);
  }
  *out = k;
  return 0;
}
extern "C" void afx_kv_destroy(afx_kv* k) {
  if (!k) return;
  (void)hipFree(k->rings); (void)hipFree(k->hist); (void)hipFree(k->feat[0]); (void)hipFree(k->feat[1]);
  delete k;
}
extern "C" size_t afx_kv_state_bytes(const afx_kv* k) {
  if (!k) return 0;
  return (size_t)k->e->cfg.n_layers * k->S * kKvSlots * 3 * kD * k->e->hsz + (size_t)k->S * kKvHist * kD * k->e->hsz + 2 * (size_t)k->S * kKvFeat * kD * 4;
}
extern "C" size_t afx_kv_workspace_bytes(const afx_kv* k, int n_frames) {
  if (!k || n_frames <= 0 || n_frames > 16) return 0;
  KvWs w;
  return kv_carve(k->e, k->S, n_frames, kKvWindow, nullptr, &w);
}
// feats6: device (S, n, 512) fp32 -- the n NEW frames of conv layer 6 (conv + LayerNorm + GELU applied: what the conv stack
// hands the feature LayerNorm), 1 <= n <= 16.  logits: device (S, 2): the back-end's logits on the window that ends with this chunk.
extern "C" int afx_kv_step(afx_kv* k, const float* feats6, int n, float* logits, void* ws, size_t ws_bytes, void* stream) {
  if (n < 1 || n > 16) return fail("afx_kv_step: a chunk brings 1..16 frames (got %d)", n);
  if (!k || !feats6 || !logits || !ws) return fail("afx_kv_step: null argument");
  afx_engine* e = k->e;
  const int S = k->S, dt = e->dt, M = S * n, Tp = kKvHist + n, group = (int)(k->hop % kKvGroups);
  const int Th = std::min(k->nfeat + n, kKvWindow);
  if (e->cfg.arch == AFX_ARCH_XLSR_AASIST && Th < 6) return fail("afx_kv_step: the AASIST head needs at least 6 frames in the window");
  KvWs w;
  const size_t needb = kv_carve(e, S, n, Th, ws, &w);
  if (ws_bytes < needb) return fail("afx_kv_step: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  hipStream_t s = (hipStream_t)stream;
  begin_call(e, nullptr);
  if (e->s3) {  // split precision: the chunk's products take pair-form operands written by their producers, or converted into the scratch
    t_s3planes = w.s3planes;
    t_s3bytes = w.s3bytes;
    s3_begin({w.feats_h, w.hbuf, w.att, w.ff, w.xpad, (char*)w.xpad + (size_t)kKvHist * kD * e->hsz});  // (+ the chunk's view of the padded rows)
  }
  const size_t hs = e->hsz;
  k->cnt[group] = n;
  // feature LayerNorm -> operand type
  {
    RowNormArgs a = plain_norm(feats6, kC, M, kC, e->F("ssl.layer_norm.weight"), e->F("ssl.layer_norm.bias"));
    a.out_h = w.feats_h; a.ldo_h = kC;
    KOK(launch_rownorm(a, dt, s));
  }
  // positional conv operand: [64 zero rows | 64 cached projected frames | the chunk | 64 zero rows] per stream
  const size_t xrow = (size_t)kD * hs, xpad_pitch = (size_t)(Tp + kPosK) * xrow;
  HIP_OK(hipMemcpy2DAsync((char*)w.xpad + kPosPad * xrow, xpad_pitch, k->hist, kKvHist * xrow, kKvHist * xrow, S, hipMemcpyDeviceToDevice, s));
  {
    GemmArgs g = plain_gemm(w.feats_h, kC, e->projw, kC, M, kD, kC);
    g.rpb = n; g.a_batch = (long)n * kC; g.a_row = kC;
    g.bias = e->F("ssl.post_extract_proj.bias");
    g.out_f = w.x; g.ldo_f = kD; g.o_batch_rows = n;  // the chunk's residual rows, dense (S, n, 1024)
    g.out_h = w.xpad; g.ldo_h = kD; g.oh_batch_rows = Tp + kPosK; g.oh_row_off = kPosPad + kKvHist;
    KOK(launch_gemm(g, dt, 1, s));
    KOK(timed(PC_MISC, 0, s, [&] { return launch_zero_pad_rows(w.xpad, S, Tp, kD, kPosPad, kPosK - kPosPad, dt, s, nullptr); }));
  }
  // The positional conv of the chunk's n frames ONLY (round 4: it used to run over all 64 + n rows of the padded layout and
  // keep the last n -- 6x the work at n = 13): output frame j reads rows [64 + j, 192 + j) of the stream's padded rows, i.e. rows
  // [j, j + 128) from the first cached frame on.
  void* xchunk = (char*)w.xpad + (size_t)kKvHist * xrow;
  if (!e->s3) {
    PosConvArgs pc;
    memset(&pc, 0, sizeof pc);
    pc.xpad = xchunk; pc.xpad_batch = (long)(Tp + kPosK) * kD; pc.W = e->posw; pc.bias = e->F("ssl.encoder.pos_conv.0.bias");
    pc.x = w.x; pc.B = S; pc.T = n;
    KOK(timed(PC_POSCONV, 2.0 * S * n * kD * (kD / kPosG) * kPosK, s, [&] { return launch_posconv(pc, dt, s); }));
  } else {  // split precision: the grouped product of run_trunk (chunked K over the time-padded pair-form rows)
    const int cpg = kD / kPosG;
    s3_set(xchunk, kS3ScaleFree);  // (history rows copied in above + the rows the projection just wrote: pair form, scale 1)
    GemmArgs g = plain_gemm(xchunk, 0, e->posw, (long)cpg * kPosK, S * n, cpg, cpg * kPosK);
    g.rpb = n; g.a_batch = (long)(Tp + kPosK) * kD; g.a_row = kD;
    g.kchunk = cpg; g.kchunk_stride = kD;
    g.g_a = cpg; g.g_w = (long)cpg * cpg * kPosK; g.g_n = cpg;
    g.bias = e->F("ssl.encoder.pos_conv.0.bias");
    g.act = ACT_GELU;
    g.resid = w.x; g.ldr = kD;
    g.out_f = w.x; g.ldo_f = kD; g.o_batch_rows = n; g.oh_batch_rows = n;
    KOK(launch_gemm(g, dt, kPosG, s));
  }
  // the newest 64 projected frames become the next chunk's left context
  HIP_OK(hipMemcpy2DAsync(k->hist, kKvHist * xrow, (char*)w.xpad + (size_t)(kPosPad + n) * xrow, xpad_pitch, kKvHist * xrow, S, hipMemcpyDeviceToDevice, s));
  for (int l = 0; l < e->cfg.n_layers; ++l) {
    const std::string P = "ssl.encoder.layers." + std::to_string(l) + ".";
    void* ring = (char*)k->rings + (size_t)l * S * kKvSlots * 3 * kD * hs;
    RowNormArgs n1 = plain_norm(w.x, kD, M, kD, e->F(P + "self_attn_layer_norm.weight"), e->F(P + "self_attn_layer_norm.bias"));
    n1.out_h = w.hbuf; n1.ldo_h = kD;
    KOK(launch_rownorm(n1, dt, s));
    // q | k | v of the chunk go straight into its 16-slot group of the ring (row remap of the epilogue): K / V are cached by being written
    GemmArgs q = plain_gemm(w.hbuf, kD, e->wqkv[l], kD, M, 3 * kD, kD);
    q.rpb = n; q.a_batch = (long)n * kD; q.a_row = kD;
    q.bias = e->bqkv[l];
    q.out_h = ring; q.ldo_h = 3 * kD; q.oh_batch_rows = kKvSlots; q.oh_row_off = group * 16;
    KOK(launch_gemm(q, dt, 1, s));
    KOK(timed(PC_MHSA, 4.0 * S * kH * 16.0 * kKvSlots * 64, s, [&] {
      if (e->s3) {  // fp32 [q | k | v] slots, hi / lo pairs split in the kernel; the output leaves as the output projection's pair-form operand
        const bool pairs = s3_ok(w.att);
        s3_set(w.att, pairs ? kS3ScaleFree : 0.f);
        return launch_mhsa_ring_split((const float*)ring, (float*)w.att, S, kH, group, k->cnt, s, pairs, kS3ScaleFree);
      }
      return launch_mhsa_ring(ring, w.att, S, kH, group, k->cnt, dt, s);
    }));
    GemmArgs o = plain_gemm(w.att, kD, e->wo[l], kD, M, kD, kD);
    o.rpb = n; o.a_batch = 16L * kD; o.a_row = kD; o.o_batch_rows = n;
    o.bias = e->F(P + "self_attn.out_proj.bias"); o.resid = w.x; o.ldr = kD; o.out_f = w.x; o.ldo_f = kD;
    KOK(launch_gemm(o, dt, 1, s));
    RowNormArgs n2 = plain_norm(w.x, kD, M, kD, e->F(P + "final_layer_norm.weight"), e->F(P + "final_layer_norm.bias"));
    n2.out_h = w.hbuf; n2.ldo_h = kD;
    KOK(launch_rownorm(n2, dt, s));
    GemmArgs f1 = plain_gemm(w.hbuf, kD, e->w1[l], kD, M, kF, kD);
    f1.bias = e->F(P + "fc1.bias"); f1.act = ACT_GELU; f1.out_h = w.ff; f1.ldo_h = kF;
    KOK(launch_gemm(f1, dt, 1, s));
    GemmArgs f2 = plain_gemm(w.ff, kF, e->w2[l], kF, M, kD, kF);
    f2.bias = e->F(P + "fc2.bias"); f2.resid = w.x; f2.ldr = kD; f2.out_f = w.x; f2.ldo_f = kD;
    KOK(launch_gemm(f2, dt, 1, s));
  }
  // final LayerNorm: the chunk's features join the right-aligned window (old frames move up by n in the other buffer)
  float *cur = k->feat[k->pp], *nxt = k->feat[k->pp ^ 1];
  const size_t frow = (size_t)kD * 4, fpitch = (size_t)kKvFeat * frow;
  HIP_OK(hipMemcpy2DAsync(nxt, fpitch, (char*)cur + (size_t)n * frow, fpitch, (size_t)(kKvFeat - n) * frow, S, hipMemcpyDeviceToDevice, s));
  {
    RowNormArgs nf = plain_norm(w.x, kD, M, kD, e->F("ssl.encoder.layer_norm.weight"), e->F("ssl.encoder.layer_norm.bias"));
    nf.out_f = nxt; nf.ldo_f = kD; nf.rpb = n; nf.o_batch_rows = kKvFeat; nf.o_row_off = kKvFeat - n;
    nf.nonfinite = e->nonfinite;
    KOK(launch_rownorm(nf, dt, s));
  }
  k->pp ^= 1;
  k->nfeat = std::min(k->nfeat + n, kKvFeat);
  // back-end on the window of the last Th frames
  HIP_OK(hipMemcpy2DAsync(w.head.ssl_f, (size_t)Th * frow, (char*)nxt + (size_t)(kKvFeat - Th) * frow, fpitch, (size_t)Th * frow, S, hipMemcpyDeviceToDevice, s));
  if (e->cfg.arch == AFX_ARCH_CONFORMER) {
    if (e->s3) {  // (fp32 operand buffers: the head's LL converts the rows it reads into its own scratch)
      HIP_OK(hipMemcpyAsync(w.head.ssl_h, w.head.ssl_f, (size_t)S * Th * kD * 4, hipMemcpyDeviceToDevice, s));
    } else {
      hipLaunchKernelGGL(f32_to_half_kernel, dim3(1024), dim3(256), 0, s, w.head.ssl_f, (uint16_t*)w.head.ssl_h, (size_t)S * Th * kD, dt == AFX_DT_BF16 ? 1 : 0);
      HIP_OK(hipGetLastError());
    }
  }
  k->hop += 1;
  if (e->taps_on && tap(e, "ssl", w.head.ssl_f, (size_t)S * Th * kD, false, s)) return 1;
  begin_call(e, &w.head);  // (the back-end's products use the back-end workspace's own scratch and buffer list)
  return run_head(e, S, Th, w.head, logits, s);
}

#undef launch_gemm
#undef launch_rownorm

static int check_call(afx_handle h, const void* in, int B, int L, const void* out, void* ws, bool needs_trunk = false) {
  if (!h || !in || !out || !ws) return fail("afx: null argument");
  if (needs_trunk && h->cfg.arch == AFX_ARCH_CONFORMER_HEAD) return fail("afx: this handle holds Conformer blocks only (MyConformer); use afx_conformer_forward");
  if (!h->finalized) return fail("afx: weights not finalized (call afx_finalize after afx_load_weight)");
  if (B <= 0 || L <= 0) return fail("afx: empty batch (B=%d, L=%d)", B, L);
  return 0;
}

extern "C" int afx_forward(afx_handle h, const float* wave, int B, int L, float* logits, void* ws, size_t ws_bytes,
                           void* stream) {
  if (check_call(h, wave, B, L, logits, ws, true)) return 1;
  Ws w;
  const size_t needb = carve(h, B, L, 0, ws, &w);
  if (ws_bytes < needb) return fail("afx_forward: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  hipStream_t s = (hipStream_t)stream;
  begin_call(h, &w);
  if (run_trunk(h, wave, B, L, w, s)) return 1;
  return run_head(h, B, w.T[6], w, logits, s);
}

// The forward in two halves, so that a scoring loop can run the back-end of batch i on a second stream under the trunk of
// batch i+1 (the AASIST head is 11 % of the teacher's time on at most 132 workgroups: alone it leaves half the chip idle):
// afx_trunk_forward leaves the SSL features in the workspace, afx_head_from_workspace runs the back-end on them.  The caller
// orders the two calls (an event between the streams) and alternates two workspaces; same kernels, same bits as afx_forward.
extern "C" int afx_trunk_forward(afx_handle h, const float* wave, int B, int L, void* ws, size_t ws_bytes, void* stream) {
  if (check_call(h, wave, B, L, ws, ws, true)) return 1;
  Ws w;
  const size_t needb = carve(h, B, L, 0, ws, &w);
  if (ws_bytes < needb) return fail("afx_trunk_forward: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  begin_call(h, &w);
  return run_trunk(h, wave, B, L, w, (hipStream_t)stream);
}
extern "C" int afx_head_from_workspace(afx_handle h, int B, int L, float* logits, void* ws, size_t ws_bytes, void* stream) {
  if (check_call(h, ws, B, L, logits, ws, true)) return 1;
  if (h->cfg.arch == AFX_ARCH_SSL) return fail("afx_head_from_workspace: this handle has no back-end");
  Ws w;
  const size_t needb = carve(h, B, L, 0, ws, &w);
  if (ws_bytes < needb) return fail("afx_head_from_workspace: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  if (w.T[6] < 1) return fail("afx_head_from_workspace: %d samples are too few for one output frame", L);
  begin_call(h, &w);
  // split precision: what afx_trunk_forward left in the workspace for a dense product of the head -- the Conformer head's LL
  // reads the features' operand copy, which the trunk's final LayerNorm wrote as pair-form rows (run_trunk: P_rownorm) -- is
  // re-registered here: the registry of which buffers hold pair-form rows lives per call, and this is a second call
  if (h->s3 && s3_ok(w.ssl_h) && (kD & 31) == 0) {
    float sc = kS3ScaleBounded;
    auto it = h->ln_scale.find(h->F("ssl.encoder.layer_norm.weight"));
    if (it != h->ln_scale.end()) sc = it->second;
    s3_set(w.ssl_h, sc);
  }
  return run_head(h, B, w.T[6], w, logits, (hipStream_t)stream);
}

// The path from the output of conv layer 5 on (conv layer 6, feature LayerNorm, projection, positional conv,
// transformer layers, head): what a streaming caller runs every hop after it has produced only the NEW frames of
// conv layers 0-5 (afx/streaming.py).  Same kernels, same order as afx_forward from that point on.
extern "C" size_t afx_tail_workspace_bytes(afx_handle h, int B, int T5) {
  if (!h || B <= 0 || T5 <= 0) return 0;
  Ws w;
  return carve(h, B, 0, 0, nullptr, &w, T5);
}
extern "C" int afx_tail_forward_strided(afx_handle h, const void* conv5_h, long batch_stride, int B, int T5, float* logits,
                                        void* ws, size_t ws_bytes, void* stream) {
  if (check_call(h, conv5_h, B, T5, logits, ws)) return 1;
  if (h->cfg.arch == AFX_ARCH_SSL || h->cfg.arch == AFX_ARCH_CONFORMER_HEAD) return fail("afx_tail_forward: this handle has no trunk + back-end");
  if (batch_stride != 0 && batch_stride < (long)T5 * kC) return fail("afx_tail_forward_strided: batch stride %ld is shorter than a window (%ld)", batch_stride, (long)T5 * kC);
  if (batch_stride % 8) return fail("afx_tail_forward_strided: the batch stride must keep rows 16-byte aligned");
  if (h->s3 && batch_stride != 0 && batch_stride != (long)T5 * kC) return fail("afx_tail_forward_strided: split precision reads a packed window");
  Ws w;
  const size_t needb = carve(h, B, 0, 0, ws, &w, T5);
  if (ws_bytes < needb) return fail("afx_tail_forward: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  if (w.T[6] < 1) return fail("afx_tail_forward: %d conv-layer-5 frames are too few for one output frame", T5);
  hipStream_t s = (hipStream_t)stream;
  begin_call(h, &w);
  if (run_trunk(h, nullptr, B, 0, w, s, conv5_h, batch_stride)) return 1;
  return run_head(h, B, w.T[6], w, logits, s);
}
extern "C" int afx_tail_forward(afx_handle h, const void* conv5_h, int B, int T5, float* logits, void* ws, size_t ws_bytes,
                                void* stream) {
  return afx_tail_forward_strided(h, conv5_h, 0, B, T5, logits, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------
// ragged batches: clips of different lengths in ONE forward, each scored exactly as if alone.
// The conv stack is local and unpadded, so frame t of a clip depends on its first 400 + 320 t samples only: a clip
// zero-padded to the batch's longest gives its own T_b frames unchanged, followed by frames nobody may look at.
// "Nobody looks" = key-padding masks (attention keys beyond T_b staged as zeros and masked to -1e30: they contribute
// exactly 0), zeros in the positional conv's operand copy and in the depthwise conv's input beyond T_b, and the
// class token / logits taken per utterance.  Everything else on the path is row-local.  The AASIST graph back-end has
// per-utterance graph sizes (T_b / 3 temporal nodes): it runs once per distinct length on the gathered sub-batch.
// ---------------------------------------------------------------------------------
static int ragged_setup(afx_engine* h, int B, int Lmax, const int* n_samples, Ws& w, std::vector<int>& frames, hipStream_t s) {
  frames.resize(B);
  for (int b = 0; b < B; ++b) {
    if (n_samples[b] > Lmax) return fail("afx_forward_ragged: clip %d has %d samples, the batch rows hold %d", b, n_samples[b], Lmax);
    int T[7];
    conv_lengths(n_samples[b], T);
    if (T[6] < 1) return fail("afx_forward_ragged: clip %d has %d samples, too few for one output frame (need >= 400)", b, n_samples[b]);
    frames[b] = T[6];
  }
  HIP_OK(hipMemcpyAsync(w.lens, frames.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
  HIP_OK(hipStreamSynchronize(s));  // `frames` is pageable host memory: the copy must have read it before it dies
  return 0;
}

extern "C" size_t afx_ragged_workspace_bytes(afx_handle h, int B, int Lmax) {
  if (!h || B <= 0 || Lmax <= 0) return 0;
  Ws w;
  return carve(h, B, Lmax, 0, nullptr, &w, 0, true);
}

extern "C" int afx_forward_ragged(afx_handle h, const float* wave, int B, int Lmax, const int* n_samples, float* logits,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (check_call(h, wave, B, Lmax, logits, ws, true)) return 1;
  if (!n_samples) return fail("afx_forward_ragged: null lengths");
  if (h->cfg.arch == AFX_ARCH_SSL) return fail("afx_forward_ragged: this handle is an SSL feature extractor; use afx_ssl_forward_ragged");
  if (h->cfg.pre_emphasis) return fail("afx_forward_ragged: engine-side pre-emphasis is not supported on ragged batches");
  if (h->cfg.extractor_mode == AFX_EXTRACTOR_GROUP_NORM) return fail("afx_forward_ragged: the group-norm extractor normalises over the whole clip; zero padding would enter its statistics");
  Ws w;
  const size_t needb = carve(h, B, Lmax, 0, ws, &w, 0, true);
  if (ws_bytes < needb) return fail("afx_forward_ragged: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  hipStream_t s = (hipStream_t)stream;
  std::vector<int> frames;
  if (ragged_setup(h, B, Lmax, n_samples, w, frames, s)) return 1;
  begin_call(h, &w);
  if (run_trunk(h, wave, B, Lmax, w, s)) return 1;
  const int Tmax = w.T[6];
  if (h->cfg.arch == AFX_ARCH_CONFORMER) return run_head(h, B, Tmax, w, logits, s);
  // AASIST: one uniform sub-batch per distinct length
  std::vector<char> done(B, 0);
  for (int b0 = 0; b0 < B; ++b0) {
    if (done[b0]) continue;
    const int t = frames[b0];
    std::vector<int> idx;
    for (int b = b0; b < B; ++b)
      if (!done[b] && frames[b] == t) { idx.push_back(b); done[b] = 1; }
    const int nb = (int)idx.size();
    for (int k = 0; k < nb; ++k)
      HIP_OK(hipMemcpyAsync(w.bucket_f + (size_t)k * t * kD, w.ssl_f + (size_t)idx[k] * Tmax * kD, (size_t)t * kD * 4,
                            hipMemcpyDeviceToDevice, s));
    if (const char* m = aasist_forward(h->aw, w.bucket_f, nb, t, w.aa, w.bucket_logits, s, h->nonfinite + 1)) return fail("%s", m);
    for (int k = 0; k < nb; ++k)
      HIP_OK(hipMemcpyAsync(logits + (size_t)idx[k] * 2, w.bucket_logits + (size_t)k * 2, 8, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// feats: device (B,Tmax,1024) fp32, rows past an utterance's own frame count are zeroed; n_frames (host int[B], may be
// null) receives the frame counts
extern "C" int afx_ssl_forward_ragged(afx_handle h, const float* wave, int B, int Lmax, const int* n_samples, float* feats,
                                      int* n_frames, void* ws, size_t ws_bytes, void* stream) {
  if (check_call(h, wave, B, Lmax, feats, ws, true)) return 1;
  if (!n_samples) return fail("afx_ssl_forward_ragged: null lengths");
  if (h->cfg.pre_emphasis) return fail("afx_ssl_forward_ragged: engine-side pre-emphasis is not supported on ragged batches");
  if (h->cfg.extractor_mode == AFX_EXTRACTOR_GROUP_NORM) return fail("afx_ssl_forward_ragged: the group-norm extractor normalises over the whole clip; zero padding would enter its statistics");
  Ws w;
  const size_t needb = carve(h, B, Lmax, 0, ws, &w, 0, true);
  if (ws_bytes < needb) return fail("afx_ssl_forward_ragged: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  hipStream_t s = (hipStream_t)stream;
  std::vector<int> frames;
  if (ragged_setup(h, B, Lmax, n_samples, w, frames, s)) return 1;
  begin_call(h, &w);
  if (run_trunk(h, wave, B, Lmax, w, s)) return 1;
  const int Tmax = w.T[6];
  HIP_OK(hipMemcpyAsync(feats, w.ssl_f, (size_t)B * Tmax * kD * 4, hipMemcpyDeviceToDevice, s));
  for (int b = 0; b < B; ++b) {
    if (frames[b] < Tmax)
      HIP_OK(hipMemsetAsync(feats + ((size_t)b * Tmax + frames[b]) * kD, 0, (size_t)(Tmax - frames[b]) * kD * 4, s));
    if (n_frames) n_frames[b] = frames[b];
  }
  return 0;
}

extern "C" int afx_ssl_forward(afx_handle h, const float* wave, int B, int L, float* feats, void* ws, size_t ws_bytes,
                               void* stream) {
  if (check_call(h, wave, B, L, feats, ws, true)) return 1;
  Ws w;
  const size_t needb = carve(h, B, L, 0, ws, &w);
  if (ws_bytes < needb) return fail("afx_ssl_forward: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  hipStream_t s = (hipStream_t)stream;
  begin_call(h, &w);
  if (run_trunk(h, wave, B, L, w, s)) return 1;
  HIP_OK(hipMemcpyAsync(feats, w.ssl_f, (size_t)B * w.T[6] * kD * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

__global__ void f32_to_half_kernel(const float* in, uint16_t* out, size_t n, int is_bf16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (is_bf16) {
      __bf16 v = (__bf16)in[i];
      memcpy(&out[i], &v, 2);
    } else {
      _Float16 v = (_Float16)in[i];
      memcpy(&out[i], &v, 2);
    }
  }
}

extern "C" int afx_head_forward(afx_handle h, const float* feats, int B, int T, float* logits, void* ws,
                                size_t ws_bytes, void* stream) {
  if (check_call(h, feats, B, T, logits, ws)) return 1;
  if (h->cfg.arch == AFX_ARCH_SSL) return fail("afx_head_forward: this handle has no back-end");
  if (h->cfg.arch == AFX_ARCH_CONFORMER_HEAD) return fail("afx_head_forward: this handle holds Conformer blocks only; use afx_conformer_forward");
  Ws w;
  memset(&w, 0, sizeof w);
  const size_t needb = carve(h, B, 0, T, ws, &w);
  if (ws_bytes < needb) return fail("afx_head_forward: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)B * T * kD;
  HIP_OK(hipMemcpyAsync(w.ssl_f, feats, n * 4, hipMemcpyDeviceToDevice, s));
  if (h->dt == AFX_DT_FP32) {
    HIP_OK(hipMemcpyAsync(w.ssl_h, feats, n * 4, hipMemcpyDeviceToDevice, s));
  } else {
    hipLaunchKernelGGL(f32_to_half_kernel, dim3(1024), dim3(256), 0, s, feats, (uint16_t*)w.ssl_h, n,
                       h->dt == AFX_DT_BF16 ? 1 : 0);
    HIP_OK(hipGetLastError());
  }
  begin_call(h, &w);
  return run_head(h, B, T, w, logits, s);
}

// MyConformer.forward alone (models/conformer_baseline.py:22-29): tokens (B,T,emb) fp32 -> class token prepended -> the
// Conformer blocks -> logits (B,2) = fc5(token 0), embedding (B,emb) = token 0.  Workspace: afx_head_workspace_bytes(h, B, T).
extern "C" int afx_conformer_forward(afx_handle h, const float* tokens, int B, int T, float* logits, float* embedding,
                                     void* ws, size_t ws_bytes, void* stream) {
  if (check_call(h, tokens, B, T, logits, ws)) return 1;
  if (h->cfg.arch != AFX_ARCH_CONFORMER && h->cfg.arch != AFX_ARCH_CONFORMER_HEAD)
    return fail("afx_conformer_forward: this handle holds no Conformer blocks");
  Ws w;
  memset(&w, 0, sizeof w);
  const size_t needb = carve(h, B, 0, T, ws, &w);
  if (ws_bytes < needb) return fail("afx_conformer_forward: workspace too small (%zu < %zu bytes)", ws_bytes, needb);
  begin_call(h, &w);
  return run_conformer(h, B, T, w, logits, (hipStream_t)stream, tokens, embedding);
}

extern "C" size_t afx_head_workspace_bytes(afx_handle h, int B, int T) {
  if (!h || B <= 0 || T <= 0) return 0;
  Ws w;
  memset(&w, 0, sizeof w);
  return carve(h, B, 0, T, nullptr, &w);
}

// ---------------------------------------------------------------------------------
// single-kernel entry points
// ---------------------------------------------------------------------------------
#define KRET(expr)                       \
  do {                                   \
    const char* m_ = (expr);             \
    return m_ ? fail("%s", m_) : 0;      \
  } while (0)

extern "C" int afx_k_gemm(int dtype, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                          const float* bias, int act, float alpha, const float* resid, long ldr, float* out_f,
                          long ldo_f, void* out_h, long ldo_h, void* stream) {
  GemmArgs g = plain_gemm(A, lda, W, ldw, M, N, K);
  g.bias = bias; g.act = act; g.alpha = alpha; g.resid = resid; g.ldr = ldr;
  g.out_f = out_f; g.ldo_f = ldo_f; g.out_h = out_h; g.ldo_h = ldo_h;
  if (dtype == DT_FP16X3) {
    // Test hook of the split-precision product: A (M,K) and W (N,K) are FP32 here; the operand forms the engine keeps
    // (weight rows [hi | lo] + row scales, built once per checkpoint; the A planes, built per product in the workspace)
    // are built per call in temporary device memory, and the call synchronises the stream before it frees them.
    hipStream_t s = (hipStream_t)stream;
    if (K % 32 || lda != K || ldw != K || ((size_t)A & 15)) return fail("afx_k_gemm(fp16x3): contiguous 16-byte aligned rows, K %% 32 == 0");
    void *wp = nullptr, *planes = nullptr;
    const long span = ((long)M * K + 31) & ~31L;
    if (hipMalloc(&wp, (size_t)N * K * 4 + (size_t)N * 4) != hipSuccess || hipMalloc(&planes, (size_t)span * 4) != hipSuccess) {
      (void)hipFree(wp);
      return fail("afx_k_gemm(fp16x3): device allocation failed");
    }
    float* sc = (float*)((char*)wp + (size_t)N * K * 4);
    const char* m = launch_pack_linear((const float*)W, N, K, K, wp, DT_FP32, s);
    if (!m) m = launch_split_weight_rows(wp, N, K, sc, s);
    if (!m) m = launch_split_pairs((const float*)A, span, planes, kS3ScaleFree, s);
    if (!m) {
      g.A = planes; g.W = wp; g.k1 = K; g.K = 2 * K; g.kchunk = 2 * K; g.a_row = 2L * K; g.ldw = 2L * K; g.pre_scale = sc; g.a_inv = 1.0f / kS3ScaleFree;
      m = launch_gemm(g, DT_FP16X3, 1, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(wp);
    (void)hipFree(planes);
    return m ? fail("%s", m) : 0;
  }
  KRET(launch_gemm(g, dtype, 1, (hipStream_t)stream));
}
extern "C" int afx_k_conv_gemm(int dtype, const void* in_h, const void* Wp, int B, int Tin, int Tout, int Cin, int k,
                               int s_, int N, const float* bias, float* out_f, void* stream) {
  GemmArgs g = plain_gemm(in_h, 0, Wp, (long)k * Cin, B * Tout, N, k * Cin);
  g.rpb = Tout; g.a_batch = (long)Tin * Cin; g.a_row = (long)s_ * Cin;
  g.o_batch_rows = Tout; g.oh_batch_rows = Tout;
  g.bias = bias; g.out_f = out_f; g.ldo_f = N;
  KRET(launch_gemm(g, dtype, 1, (hipStream_t)stream));
}
extern "C" int afx_k_conv_ln_act(int dtype, const void* in_h, const void* Wp, int B, int Tin, int Tout, int Cin,
                                 int k, int s_, const float* bias, const float* gamma, const float* beta, float eps,
                                 int act, float* out_f, void* out_h, void* stream) {
  GemmArgs g = plain_gemm(in_h, 0, Wp, (long)k * Cin, B * Tout, 512, k * Cin);
  g.rpb = Tout; g.a_batch = (long)Tin * Cin; g.a_row = (long)s_ * Cin;
  g.o_batch_rows = Tout; g.oh_batch_rows = Tout;
  g.bias = bias; g.act = act; g.ln_gamma = gamma; g.ln_beta = beta; g.ln_eps = eps;
  g.out_f = out_f; g.ldo_f = 512; g.out_h = out_h; g.ldo_h = 512;
  KRET(launch_gemm(g, dtype, 1, (hipStream_t)stream));
}
extern "C" int afx_k_pack_linear(int dtype, const float* w, int N, int K, int Kpad, void* out_h, void* stream) {
  KRET(launch_pack_linear(w, N, K, Kpad, out_h, dtype, (hipStream_t)stream));
}
extern "C" int afx_k_pack_conv(int dtype, const float* w, int N, int Cin, int k, void* out_h, void* stream) {
  KRET(launch_pack_conv(w, N, Cin, k, out_h, dtype, (hipStream_t)stream));
}
extern "C" size_t afx_k_conv0_pack_bytes(void) { return conv0_pack_bytes(); }
extern "C" int afx_k_conv0_pack(const float* w, const float* bias, void* pack, void* stream) {
  if (!w || !bias || !pack) return fail("afx_k_conv0_pack: null argument");
  KRET(launch_conv0_pack(w, bias, pack, (hipStream_t)stream));
}
// pack: the layer's split-precision operand block (afx_k_conv0_pack, built ONCE per checkpoint); asynchronous, allocates nothing
extern "C" int afx_k_conv0_packed(int dtype, const float* wave, int B, int L, const void* pack, const float* w,
                                  const float* bias, const float* gamma, const float* beta, int pre_emph, float coef,
                                  void* out_h, void* stream) {
  if (dtype != DT_FP32 && !pack) return fail("afx_k_conv0_packed: null pack");
  KRET(launch_conv0(wave, B, L, (L - 10) / 5 + 1, w, bias, gamma, beta, pre_emph, coef, out_h, dtype, (hipStream_t)stream,
                    dtype != DT_FP32 ? pack : nullptr));
}
// (test hook of round 1: builds the pack per call -- a device allocation and a stream synchronisation; hot paths use
// afx_k_conv0_pack once + afx_k_conv0_packed)
extern "C" int afx_k_conv0(int dtype, const float* wave, int B, int L, const float* w, const float* bias,
                           const float* gamma, const float* beta, int pre_emph, float coef, void* out_h,
                           void* stream) {
  void* pack = nullptr;
  if (dtype != DT_FP32) {
    if (hipMalloc(&pack, conv0_pack_bytes()) != hipSuccess) return fail("afx_k_conv0: device allocation failed");
    if (const char* m = launch_conv0_pack(w, bias, pack, (hipStream_t)stream)) {
      (void)hipFree(pack);
      return fail("%s", m);
    }
  }
  const char* m = launch_conv0(wave, B, L, (L - 10) / 5 + 1, w, bias, gamma, beta, pre_emph, coef, out_h, dtype,
                               (hipStream_t)stream, pack);
  if (pack) {
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(pack);
  }
  if (m) return fail("%s", m);
  return 0;
}
extern "C" int afx_debug_set(const char* key, int value) {
  if (!key) return fail("afx_debug_set: null key");
  if (!strcmp(key, "gemm_map")) {
    gemm_set_map_mode(value);
    return 0;
  }
  if (!strcmp(key, "aasist_stop")) {
    aasist_set_stop(value);
    return 0;
  }
  if (!strcmp(key, "gemm_tile")) {
    gemm_set_tile(value);
    return 0;
  }
  if (!strcmp(key, "gemm_small_deep")) {
    gemm_set_small_deep(value);
    return 0;
  }
  if (!strcmp(key, "gemm_s3_small")) {
    gemm_set_s3_small(value);
    return 0;
  }
  if (!strcmp(key, "gemm_split")) {
    gemm_set_split(value);
    return 0;
  }
  if (!strcmp(key, "gemm_conv_split")) {
    gemm_set_conv_split(value);
    return 0;
  }
  if (!strcmp(key, "gemm_ph4")) {
    gemm_set_ph4(value);
    return 0;
  }
  if (!strcmp(key, "gemm_fit")) {
    gemm_set_fit(value);
    return 0;
  }
  if (!strcmp(key, "conf_attn_waves")) {
    conf_attn_mfma_set_waves(value);
    return 0;
  }
  if (!strcmp(key, "mhsa_vtr")) {
    mhsa_set_vtr(value);
    return 0;
  }
  if (!strcmp(key, "mhsa_zsplit")) {
    mhsa_set_zsplit(value);
    return 0;
  }
  if (!strcmp(key, "mhsa_waves")) {
    mhsa_set_waves(value);
    return 0;
  }
  if (!strcmp(key, "mhsa_force_long")) {
    mhsa_set_force_long(value);
    return 0;
  }
  if (!strcmp(key, "conf_attn_force_long")) {
    conf_attn_mfma_set_force_long(value);
    return 0;
  }
  if (!strcmp(key, "conf_attn_block")) {
    conf_attn_set_block(value);
    return 0;
  }
  if (!strcmp(key, "gemm_deep")) {
    gemm_set_deep(value);
    return 0;
  }
  if (!strcmp(key, "gemm_nodma")) {
    if (!gemm_set_nodma(value)) return fail("afx_debug_set: gemm_nodma exists only in the attribution build (make attr, AFX_LIB=.../libafx_attr.so)");
    return 0;
  }
  if (!strcmp(key, "gemm_a_nt")) {
    gemm_set_a_nt(value);
    return 0;
  }
  if (!strcmp(key, "conv0_mfma")) {
    conv0_set_mfma(value);
    return 0;
  }

  return fail("afx_debug_set: unknown key '%s'", key);
}
extern "C" int afx_engine_set(afx_handle h, const char* key, int value) {
  if (!h || !key) return fail("afx_engine_set: null argument");
  if (!strcmp(key, "posconv_sliding")) h->posconv_sliding = value != 0;
  else if (!strcmp(key, "conf_attn_mfma")) h->conf_attn_mfma = value != 0;
  else if (!strcmp(key, "fuse_conformer")) h->fuse_conformer = value != 0;
  else if (!strcmp(key, "fuse_conv_ln")) h->fuse_conv_ln = value != 0;
  else if (!strcmp(key, "gemm_small_deep")) h->gemm_small_deep = value != 0;
  else return fail("afx_engine_set: unknown key '%s'", key);
  return 0;
}
extern "C" int afx_k_pre_emphasis(const float* x, int B, int L, float coef, float* y, void* stream) {
  KRET(launch_pre_emphasis(x, B, L, coef, y, (hipStream_t)stream));
}
extern "C" int afx_k_tile_crop(const float* x, const long long* offs, const long long* starts, int B, int duration,
                               float* out, void* stream) {
  if (!x || !offs || !out) return fail("afx_k_tile_crop: null argument");
  KRET(launch_tile_crop(x, offs, starts, B, duration, out, (hipStream_t)stream));
}
extern "C" int afx_k_rownorm(int dtype, const float* x, long ldx, int rows, int C, const float* gamma,
                             const float* beta, float eps, int act, float* out_f, long ldo_f, void* out_h, long ldo_h,
                             void* stream) {
  RowNormArgs a = plain_norm(x, ldx, rows, C, gamma, beta);
  a.eps = eps; a.act = act; a.out_f = out_f; a.ldo_f = ldo_f; a.out_h = out_h; a.ldo_h = ldo_h;
  KRET(launch_rownorm(a, dtype, (hipStream_t)stream));
}
extern "C" int afx_k_mhsa(int dtype, const void* qkv, void* out, int B, int T, int H, void* stream) {
  if (dtype == DT_FP16X3) KRET(launch_mhsa_split((const float*)qkv, (float*)out, B, T, H, (hipStream_t)stream));  // fp32 rows in / out
  KRET(launch_mhsa(qkv, out, B, T, H, dtype, (hipStream_t)stream));
}
extern "C" int afx_k_conf_attn(int dtype, const float* q, long ldq, const float* kv, long ldkv, const float* rel,
                               int max_pos, int B, int N, int H, int dh, void* out_h, long ldo, void* stream) {
  KRET(launch_conf_attn(q, ldq, kv, ldkv, rel, max_pos, B, N, H, dh, out_h, ldo, dtype, (hipStream_t)stream));
}
extern "C" int afx_k_conf_attn_mfma(int dtype, const float* q, long ldq, const float* kv, long ldkv, const void* rel_h,
                                    int max_pos, int B, int N, int H, int dh, void* out_h, long ldo, void* stream) {
  if (dtype == DT_FP16X3)  // split precision: fp32 rows out, the table as fp32 rows padded to 64 (afx_k_pack_linear with dtype fp32)
    KRET(launch_conf_attn_split(q, ldq, kv, ldkv, (const float*)rel_h, max_pos, B, N, H, dh, (float*)out_h, ldo, (hipStream_t)stream));
  KRET(launch_conf_attn_mfma(q, ldq, kv, ldkv, rel_h, max_pos, B, N, H, dh, out_h, ldo, dtype, (hipStream_t)stream));
}
extern "C" int afx_k_conf_dwconv(int dtype, const float* x, long ldx, const float* w, const float* bias,
                                 const float* bn_scale, const float* bn_shift, int B, int N, int C, int k, void* out_h,
                                 long ldo, void* stream) {
  KRET(launch_conf_dwconv(x, ldx, w, bias, bn_scale, bn_shift, B, N, C, k, out_h, ldo, dtype, (hipStream_t)stream));
}
