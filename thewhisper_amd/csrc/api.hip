// C ABI (include/thewhisper.h) + context + kernel orchestration for the MI355X Whisper hot path.
// Host-side control only: every FLOP/byte of the path runs in the gfx950 kernels of k_*.hip.
#include "../../include/thewhisper.h"
#include "tw_common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <chrono>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct LayerW {
  void *ln1_g = nullptr, *ln1_b = nullptr;  // self_attn_layer_norm
  void *wqkv = nullptr, *bqkv = nullptr;    // [3d, d], [3d] (q rows pre-scaled by 1/8, k bias = 0)
  // MXFP8 contexts: block scales of the six decoder projections (inside the weight buffers, behind the fp8 fragments)
  unsigned char *s_qkv = nullptr, *s_o = nullptr, *s_qc = nullptr, *s_oc = nullptr, *s_1 = nullptr, *s_2 = nullptr;
  void *wo = nullptr, *bo = nullptr;
  void *lnx_g = nullptr, *lnx_b = nullptr;  // encoder_attn_layer_norm (decoder)
  void *wq_c = nullptr, *bq_c = nullptr;    // cross q (pre-scaled)
  void *wkv_c = nullptr, *bkv_c = nullptr;  // [2d, d] cross k|v
  void *wo_c = nullptr, *bo_c = nullptr;
  void *ln2_g = nullptr, *ln2_b = nullptr;  // final_layer_norm
  void *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
  // folded pre-LayerNorm companions of the decoder's LN'd projections (float32 [N]): see k_decode.hip
  float *qkv_gw = nullptr, *qkv_cb = nullptr, *qc_gw = nullptr, *qc_cb = nullptr, *fc1_gw = nullptr, *fc1_cb = nullptr;
  // weight rows per workgroup tile each decoder projection was laid out for (k_decode.hip: 16, or 8 / 4 for the narrow ones)
  int tr_qkv = 16, tr_o = 16, tr_qc = 16, tr_oc = 16, tr_1 = 16, tr_2 = 16;
};

}  // namespace

struct tw_ctx {
  tw_config cfg{};
  int dtype = 1;
  bool w8 = false;   // decoder projection weights in MXFP8 (TW_BF16_MXFP8 / TW_BF16_W8A16 contexts)
  int a16 = 0;       // ... contracted as bf16 after widening in registers (TW_BF16_W8A16; TW_FP8_ACT=bf16|fp8 overrides at run time for A/B)
  // "cross query ahead" (decode_core): the decoder's cross-attention query projection is folded into the two launches before
  // it, which saves one dependent launch per layer and step.  du = float32 [Bmax][d] pre-activation of that projection.
  bool fuse_cq = false;
  bool enc_fold = true;   // encoder pre-LayerNorms folded into the QKV / fc1 GEMMs (GemmEpilogue::stats_*); TW_ENC_FOLD_LN=0: two LayerNorm launches per layer
  float *ln_stats_a = nullptr, *ln_stats_b = nullptr;   // [rows][d / 32][2] partial row statistics of xa / xb
  float* du = nullptr;
  float* dstats = nullptr;   // [Bmax][d / tile rows][2] partial LayerNorm statistics of the residual stream (see GemvArgs::stats)
  unsigned char* logit_ws = nullptr;  // block scales of logit_w
  size_t esz = 2;
  int d = 0, H = 0, ffn = 0, V = 0, T = 0, Tp = 0, P = 0, C = 0, Bmax = 0, Le = 0, Ld = 0, n_mels = 0, Ha = 0;
  std::string err;
  std::vector<void*> allocs;
  hipStream_t own_stream = nullptr;

  // weights
  void *conv1_w = nullptr, *conv1_b = nullptr, *conv2_w = nullptr, *conv2_b = nullptr;
  void *enc_pos_raw = nullptr, *enc_pos = nullptr, *enc_ln_g = nullptr, *enc_ln_b = nullptr;
  void *tok_emb = nullptr, *dec_pos = nullptr, *dec_ln_g = nullptr, *dec_ln_b = nullptr;
  void* logit_w = nullptr;  // tok_emb with the final LayerNorm gain folded in (the lookup table itself stays unscaled)
  float *logit_gw = nullptr, *logit_cb = nullptr;
  std::vector<LayerW> enc, dec;
  std::set<std::string> loaded;
  bool finalized = false;
  bool shares_weights = false;   // tw_create_sibling: the weight pointers belong to another context (never freed here)

  // log-mel
  LogmelTables lm{};
  float* logspec_ws = nullptr; size_t logspec_cap = 0;
  unsigned* lm_max = nullptr;
  int* n_valid_dev = nullptr;

  // encoder workspace
  void *melT = nullptr, *h1 = nullptr, *xa = nullptr, *xb = nullptr, *lnbuf = nullptr, *qbuf = nullptr, *kbuf = nullptr,
       *vtbuf = nullptr, *attn = nullptr, *ffnh = nullptr, *enc_out = nullptr;
  int encoded_B = 0, cross_B = 0;

  // decoder state
  void *self_k = nullptr, *self_v = nullptr, *cross_k = nullptr, *cross_v = nullptr;
  unsigned char *cross_ksc = nullptr, *cross_vsc = nullptr;  // MXFP8 contexts: per-key scale bytes of the fp8 cross K / V^T caches [Ld][B][H][Tp]
  void *dx0 = nullptr, *dx1 = nullptr, *dq = nullptr, *datt = nullptr, *dh = nullptr;
  float* logits = nullptr;
  float* align = nullptr;
  int* align_slot = nullptr;  // [Ld][H]
  int *seq = nullptr, *cur_ids = nullptr, *finished = nullptr, *last_ts = nullptr;
  DecState* stt = nullptr;
  int* begin_suppress_dev = nullptr; int* suppress_dev = nullptr;
  SamplerPartial* sampler_partials = nullptr;
  unsigned* suppress_bits = nullptr;  // [(V+31)/32] static suppress list as a bitmap, rebuilt per generate call
  int* h_pinned = nullptr;  // pinned host scratch: finished ring [8][64] | n_valid [64] | DecState upload [16]
  int* row_ids = nullptr;   // device token table of a prefill (position-major), row_ids_cap ints
  int* h_rows = nullptr;    // its pinned host staging: [2][row_ids_cap] (token table | last-timestamp table of a verify round)
  // draft-and-verify (tw_greedy_opts::n_draft): per-row sampler state and results (SamplerArgs, tw_common.h)
  int* row_lastts = nullptr;     // [row_ids_cap] position-major like row_ids
  int* row_choice = nullptr;     // [row_ids_cap]
  int* verify_state = nullptr;   // [2 + 64] device
  int* h_verify = nullptr;       // pinned: [2 + 64] read back | [1] reset value
  int last_draft[4] = {0, 0, 0, 0};   // of the last greedy call: tokens offered, accepted, verify launches, verify rounds
  size_t row_ids_cap = 0;
  int row_cap = 0;          // rows the per-token activation buffers hold (>= max_batch; 64 for the prefill launches)
  int* h_stage = nullptr;   // pinned staging of tw_generate_greedy: token table [Bmax][P] (up and down) | 3 x [Bmax] | suppress lists | DecState
  size_t h_stage_ints = 0;
  hipEvent_t ring_ev[8]{};
  int last_seq_len = 0, last_n_prompt = 0;
  int dec_key_bound = 0;  // upper bound of decoder positions for the current decode (prompt + max new tokens)
  int enc_pos_rows = 0;  // rows of the encoder positional table as loaded (1500 upstream)

  // dtw workspace
  float *zbuf = nullptr, *mat = nullptr, *out_ts = nullptr;
  signed char* trace = nullptr;
  int* n_cols = nullptr;

  // graph replay of one decode step
  // one captured step per 64-key bucket of the self-attention length (the step at position s only has to read keys
  // [0, roundup(s + 1, 64)), and the host knows s): fewer bytes per step than one graph sized for the longest sequence
  std::map<std::string, hipGraphExec_t> step_graphs;
  std::string step_graph_key;

  // timing
  hipEvent_t ev0[5]{}, ev1[5]{};
  bool ev_valid[5]{};
  int last_steps = 0;
};

namespace {

int fail(tw_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(c, expr)                                                                       \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail((c), TW_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

template <typename P>
int dalloc(tw_ctx* c, P** p, size_t bytes, bool zero) {
  void* q = nullptr;
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) return fail(c, TW_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  c->allocs.push_back(q);
  if (zero) {
    e = hipMemset(q, 0, bytes);
    if (e != hipSuccess) return fail(c, TW_EHIP, "hipMemset failed: %s", hipGetErrorString(e));
  }
  *p = reinterpret_cast<P*>(q);
  return TW_OK;
}
#define ALLOC(c, ptr, bytes, zero)                   \
  do {                                               \
    int _r = dalloc((c), &(ptr), (bytes), (zero));   \
    if (_r != TW_OK) return _r;                      \
  } while (0)

inline hipStream_t pick_stream(tw_ctx* c, void* s) { return s ? reinterpret_cast<hipStream_t>(s) : c->own_stream; }

// Every entry point runs with the context's device current and restores the caller's device on exit: worker threads
// (BatchingHub, the gateway's thread pool, the reference scheduler's thread) start on device 0, and kernel launches,
// memcpys and hipMalloc go to the calling thread's CURRENT device, not to the device a stream belongs to.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define TW_ON_DEVICE(c) DeviceGuard _tw_guard((c)->cfg.device)

char* at(void* base, size_t elems, size_t esz) { return reinterpret_cast<char*>(base) + elems * esz; }

// slaney mel bank, float64 (HF:audio_utils.py:638-729), then cast to float32 as the reference does
void build_mel_bank(int n_mels, std::vector<float>& bank, std::vector<int>& lo, std::vector<int>& hi) {
  const int nb = 201;
  auto hz2mel = [](double f) { return f >= 1000.0 ? 15.0 + std::log(f / 1000.0) * (27.0 / std::log(6.4)) : 3.0 * f / 200.0; };
  auto mel2hz = [](double m) { return m >= 15.0 ? 1000.0 * std::exp((std::log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; };
  const double mmin = hz2mel(0.0), mmax = hz2mel(8000.0);
  std::vector<double> ff(n_mels + 2);
  for (int i = 0; i < n_mels + 2; ++i) ff[i] = mel2hz(mmin + (mmax - mmin) * (double)i / (double)(n_mels + 1));
  bank.assign((size_t)nb * n_mels, 0.f);
  lo.assign(n_mels, nb);
  hi.assign(n_mels, 0);
  for (int k = 0; k < nb; ++k) {
    const double fk = 8000.0 * (double)k / (double)(nb - 1);
    for (int m = 0; m < n_mels; ++m) {
      const double down = -(ff[m] - fk) / (ff[m + 1] - ff[m]);
      const double up = (ff[m + 2] - fk) / (ff[m + 2] - ff[m + 1]);
      double v = std::fmax(0.0, std::fmin(down, up));
      v *= 2.0 / (ff[m + 2] - ff[m]);
      const float vf = (float)v;
      bank[(size_t)k * n_mels + m] = vf;
      if (vf != 0.f) {
        if (k < lo[m]) lo[m] = k;
        if (k + 1 > hi[m]) hi[m] = k + 1;
      }
    }
  }
  for (int m = 0; m < n_mels; ++m)
    if (hi[m] == 0) lo[m] = 0;
}

int upload(tw_ctx* c, void* dst, const void* src, size_t bytes) {
  HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return TW_OK;
}

void tic(tw_ctx* c, int i, hipStream_t st) { (void)hipEventRecord(c->ev0[i], st); }
void toc(tw_ctx* c, int i, hipStream_t st) { (void)hipEventRecord(c->ev1[i], st); c->ev_valid[i] = true; }

}  // namespace

extern "C" {

const char* tw_version(void) { return TW_VERSION_STRING; }

const char* tw_last_error(const tw_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int tw_destroy(tw_ctx* c) {
  if (!c) return TW_OK;
  TW_ON_DEVICE(c);
  (void)hipDeviceSynchronize();
  for (auto& kv : c->step_graphs) (void)hipGraphExecDestroy(kv.second);
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_rows) (void)hipHostFree(c->h_rows);
  if (c->h_verify) (void)hipHostFree(c->h_verify);
  for (int i = 0; i < 5; ++i) {
    if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]);
    if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
  }
  for (int i = 0; i < 8; ++i)
    if (c->ring_ev[i]) (void)hipEventDestroy(c->ring_ev[i]);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return TW_OK;
}

static int create_ctx(const tw_config* cfg, const tw_ctx* share, tw_ctx** out);

int tw_create(const tw_config* cfg, tw_ctx** out) { return create_ctx(cfg, nullptr, out); }

int tw_create_sibling(const tw_ctx* src, int32_t max_batch, tw_ctx** out) {
  if (!src || !out) return fail(nullptr, TW_EINVAL, "tw_create_sibling: null argument");
  if (!src->finalized) return fail(nullptr, TW_ESTATE, "tw_create_sibling: the source context's weights are not finalized");
  if (src->shares_weights) return fail(nullptr, TW_EINVAL, "tw_create_sibling: the source is itself a sibling (create it from the owner)");
  tw_config cfg = src->cfg;
  cfg.max_batch = max_batch;
  return create_ctx(&cfg, src, out);
}

// share != nullptr: every weight pointer is taken from `share` (finalized; it owns them and must outlive this context) instead
// of being allocated; workspace, arenas, streams, events and graphs are this context's own
static int create_ctx(const tw_config* cfg, const tw_ctx* share, tw_ctx** out) {
  if (!cfg || !out) return fail(nullptr, TW_EINVAL, "tw_create: null argument");
  *out = nullptr;
  if (cfg->heads <= 0 || cfg->d_model != cfg->heads * 64)
    return fail(nullptr, TW_EINVAL, "head_dim must be 64 (d_model=%d heads=%d)", cfg->d_model, cfg->heads);
  if (cfg->dtype != TW_BF16 && cfg->dtype != TW_F32 && cfg->dtype != TW_F16 && cfg->dtype != TW_BF16_MXFP8 && cfg->dtype != TW_BF16_W8A16)
    return fail(nullptr, TW_EINVAL, "dtype must be TW_BF16, TW_F16, TW_F32, TW_BF16_MXFP8 or TW_BF16_W8A16");
  const bool cfg_w8 = cfg->dtype == TW_BF16_MXFP8 || cfg->dtype == TW_BF16_W8A16;
  if (cfg_w8 && (cfg->d_model % 128 || cfg->ffn % 128))
    return fail(nullptr, TW_EINVAL, "TW_BF16_MXFP8 / TW_BF16_W8A16 need d_model and ffn to be multiples of 128");
  if (cfg->d_model % 64 || cfg->ffn % 64 || cfg->d_model > 1280)
    return fail(nullptr, TW_EINVAL, "d_model/ffn must be multiples of 64 and d_model <= 1280");
  if (cfg->max_batch < 1 || cfg->max_batch > 64) return fail(nullptr, TW_EINVAL, "max_batch must be in [1,64]");
  if (cfg->dtype == TW_BF16_MXFP8 && cfg->max_batch > 16)
    return fail(nullptr, TW_EINVAL, "TW_BF16_MXFP8: max_batch must be in [1,16] (TW_BF16_W8A16 takes up to 64 streams)");
  if (cfg->source_positions < 8 || cfg->source_positions > 1500 || cfg->target_positions < 8 || cfg->target_positions > 511)
    return fail(nullptr, TW_EINVAL, "source_positions must be in [8,1500], target_positions in [8,511]");
  if (cfg->n_align_heads < 0 || cfg->n_align_heads > TW_MAX_ALIGN_HEADS) return fail(nullptr, TW_EINVAL, "bad n_align_heads");
  if (cfg->n_mels < 1 || cfg->n_mels > 256) return fail(nullptr, TW_EINVAL, "bad n_mels");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, TW_EHIP, "no HIP device available (the MI355X path has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, TW_EINVAL, "device %d out of range", cfg->device);
  DeviceGuard guard(cfg->device);   // allocations below land on the context's device; the caller's device is restored on return
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device) return fail(nullptr, TW_EHIP, "hipSetDevice(%d) failed", cfg->device);

  tw_ctx* c = new tw_ctx();
  c->cfg = *cfg;
  c->w8 = cfg_w8;
  c->a16 = cfg->dtype == TW_BF16_W8A16;
  if (cfg->dtype == TW_BF16_MXFP8) {   // same weights and layouts: the activation flavour can be switched for A/B measurements
    const char* fa = getenv("TW_FP8_ACT");
    if (fa && !strcmp(fa, "bf16")) c->a16 = 1;
  }
  c->dtype = c->w8 ? (int)TW_BF16 : cfg->dtype;
  c->esz = c->dtype == TW_F32 ? 4 : 2;
  c->d = cfg->d_model; c->H = cfg->heads; c->ffn = cfg->ffn; c->V = cfg->vocab;
  c->T = cfg->source_positions; c->Tp = (c->T + 63) / 64 * 64; c->P = cfg->target_positions;
  c->n_mels = cfg->n_mels; c->C = (cfg->n_mels + 63) / 64 * 64;
  c->Bmax = cfg->max_batch; c->Le = cfg->enc_layers; c->Ld = cfg->dec_layers; c->Ha = cfg->n_align_heads;
  {
    const char* fe = getenv("TW_FUSE_CQ_LAYOUT");   // 0: do not even lay the weights out for the fused sequence
    c->fuse_cq = c->d == c->H * 64 && !(fe && atoi(fe) == 0);
    const char* ef = getenv("TW_ENC_FOLD_LN");
    c->enc_fold = c->d % 64 == 0 && c->d <= 1280 && !(ef && atoi(ef) == 0);   // (k_gemm.hip: TW_LN_HALF partial statistics per thread)
  }
  auto bail = [&](int r) { g_create_error = c->err; tw_destroy(c); return r; };
#define CALLOC(ptr, bytes, zero) do { int _r = dalloc(c, &(ptr), (bytes), (zero)); if (_r != TW_OK) return bail(_r); } while (0)
#define CHIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fail(c, TW_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); return bail(TW_EHIP); } } while (0)
  CHIP(init_decode_kernels());
  CHIP(hipStreamCreate(&c->own_stream));
  for (int i = 0; i < 5; ++i) { CHIP(hipEventCreate(&c->ev0[i])); CHIP(hipEventCreate(&c->ev1[i])); }
  for (int i = 0; i < 8; ++i) CHIP(hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming));
  CHIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), sizeof(int) * (8 * 64 + 64 + 16), hipHostMallocDefault));
  c->h_stage_ints = (size_t)c->Bmax * c->P + 3 * (size_t)c->Bmax + 64 + 1024 + 16;
  CHIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), sizeof(int) * c->h_stage_ints, hipHostMallocDefault));

  const size_t e = c->esz;
  const size_t d = c->d, H = c->H, F = c->ffn, V = c->V, T = c->T, Tp = c->Tp, P = c->P, C = c->C, B = c->Bmax;
  if (share) {
    c->shares_weights = true;
    c->fuse_cq = share->fuse_cq;
    c->enc_fold = share->enc_fold;
    c->a16 = share->a16;
    c->conv1_w = share->conv1_w; c->conv1_b = share->conv1_b; c->conv2_w = share->conv2_w; c->conv2_b = share->conv2_b;
    c->enc_pos_raw = share->enc_pos_raw; c->enc_pos = share->enc_pos; c->enc_ln_g = share->enc_ln_g; c->enc_ln_b = share->enc_ln_b;
    c->tok_emb = share->tok_emb; c->dec_pos = share->dec_pos; c->dec_ln_g = share->dec_ln_g; c->dec_ln_b = share->dec_ln_b;
    c->logit_w = share->logit_w; c->logit_gw = share->logit_gw; c->logit_cb = share->logit_cb; c->logit_ws = share->logit_ws;
    c->enc = share->enc; c->dec = share->dec;
    c->loaded = share->loaded; c->enc_pos_rows = share->enc_pos_rows; c->finalized = true;
  } else {
  // ---- weights ----
  CALLOC(c->conv1_w, d * 3 * C * e, true);
  CALLOC(c->conv1_b, d * e, true);
  CALLOC(c->conv2_w, d * 3 * d * e, true);
  CALLOC(c->conv2_b, d * e, true);
  CALLOC(c->enc_pos_raw, (size_t)1500 * d * 4, true);
  CALLOC(c->enc_pos, T * d * e, true);
  CALLOC(c->enc_ln_g, d * e, true);
  CALLOC(c->enc_ln_b, d * e, true);
  CALLOC(c->tok_emb, V * d * e, true);
  CALLOC(c->dec_pos, P * d * e, true);
  CALLOC(c->dec_ln_g, d * e, true);
  CALLOC(c->dec_ln_b, d * e, true);
  CALLOC(c->logit_w, ((V + 15) / 16 * 16) * d * e, false);  // padded to whole 16-row tiles
  CALLOC(c->logit_gw, V * 4, true);
  CALLOC(c->logit_cb, V * 4, true);
  c->enc.resize(c->Le);
  c->dec.resize(c->Ld);
  auto alloc_layer = [&](LayerW& L, bool decoder) -> int {
    int r;
#define LA(ptr, n) if ((r = dalloc(c, &(ptr), (n) * e, true)) != TW_OK) return r
    // decoder with "cross query ahead": a fourth block of rows behind q|k|v (the folded cross-query weight) and a second one
    // behind the out-projection (cross-query weight . out-projection), see tw_finalize_weights
    const size_t qkv_blocks = (decoder && c->fuse_cq) ? 4 : 3, o_blocks = (decoder && c->fuse_cq) ? 2 : 1;
    LA(L.ln1_g, d); LA(L.ln1_b, d); LA(L.wqkv, qkv_blocks * d * d); LA(L.bqkv, 3 * d); LA(L.wo, o_blocks * d * d); LA(L.bo, d);
    LA(L.ln2_g, d); LA(L.ln2_b, d); LA(L.w1, F * d); LA(L.b1, F); LA(L.w2, d * F); LA(L.b2, d);
    if (!decoder && c->enc_fold) {
      if ((r = dalloc(c, &L.qkv_gw, 3 * d * 4, true)) != TW_OK) return r;
      if ((r = dalloc(c, &L.qkv_cb, 3 * d * 4, true)) != TW_OK) return r;
      if ((r = dalloc(c, &L.fc1_gw, F * 4, true)) != TW_OK) return r;
      if ((r = dalloc(c, &L.fc1_cb, F * 4, true)) != TW_OK) return r;
    }
    if (decoder) {
      if ((r = dalloc(c, &L.qkv_gw, 4 * d * 4, true)) != TW_OK) return r;
      if ((r = dalloc(c, &L.qkv_cb, 4 * d * 4, true)) != TW_OK) return r;
      if ((r = dalloc(c, &L.qc_gw, d * 4, true)) != TW_OK) return r;
      if ((r = dalloc(c, &L.qc_cb, d * 4, true)) != TW_OK) return r;
      if ((r = dalloc(c, &L.fc1_gw, F * 4, true)) != TW_OK) return r;
      if ((r = dalloc(c, &L.fc1_cb, F * 4, true)) != TW_OK) return r;
      LA(L.lnx_g, d); LA(L.lnx_b, d); LA(L.wq_c, d * d); LA(L.bq_c, d); LA(L.wkv_c, 2 * d * d); LA(L.bkv_c, 2 * d);
      LA(L.wo_c, d * d); LA(L.bo_c, d);
    }
#undef LA
    return TW_OK;
  };
  for (auto& L : c->enc) { int r = alloc_layer(L, false); if (r != TW_OK) return bail(r); }
  for (auto& L : c->dec) { int r = alloc_layer(L, true); if (r != TW_OK) return bail(r); }
  }  // !share

  // ---- log-mel tables ----
  {
    std::vector<double> tw(400), win(400);
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < 400; ++j) { tw[j] = std::cos(two_pi * j / 400.0); win[j] = 0.5 - 0.5 * std::cos(two_pi * j / 400.0); }
    std::vector<float> bank; std::vector<int> lo, hi;
    build_mel_bank(c->n_mels, bank, lo, hi);
    double *dtw_ = nullptr, *dwin = nullptr; float* dbank = nullptr; int *dlo = nullptr, *dhi = nullptr;
    CALLOC(dtw_, 400 * 8, false); CALLOC(dwin, 400 * 8, false); CALLOC(dbank, bank.size() * 4, false);
    CALLOC(dlo, lo.size() * 4, false); CALLOC(dhi, hi.size() * 4, false);
    CHIP(hipMemcpy(dtw_, tw.data(), 400 * 8, hipMemcpyHostToDevice));
    CHIP(hipMemcpy(dwin, win.data(), 400 * 8, hipMemcpyHostToDevice));
    CHIP(hipMemcpy(dbank, bank.data(), bank.size() * 4, hipMemcpyHostToDevice));
    CHIP(hipMemcpy(dlo, lo.data(), lo.size() * 4, hipMemcpyHostToDevice));
    CHIP(hipMemcpy(dhi, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    c->lm = LogmelTables{dtw_, dwin, dbank, dlo, dhi};
    c->logspec_cap = B * (size_t)c->n_mels * (2 * T);
    CALLOC(c->logspec_ws, c->logspec_cap * 4, false);
    CALLOC(c->lm_max, B * 4, true);
    CALLOC(c->n_valid_dev, B * 4, true);
  }
  // ---- encoder workspace ----
  CALLOC(c->melT, B * (2 * T + 2) * C * e, true);
  CALLOC(c->h1, B * (2 * T + 2) * d * e, true);   // pad rows 0 and 2T+1 of every clip stay zero
  CALLOC(c->xa, B * T * d * e, false);
  CALLOC(c->xb, B * T * d * e, false);
  CALLOC(c->lnbuf, B * T * d * e, false);
  CALLOC(c->ln_stats_a, B * T * (d / 32) * 8, false);
  CALLOC(c->ln_stats_b, B * T * (d / 32) * 8, false);
  CALLOC(c->qbuf, B * T * d * e, false);
  CALLOC(c->kbuf, B * T * d * e, false);
  CALLOC(c->vtbuf, B * H * 64 * Tp * e, true);      // key padding [T, Tp) stays zero
  CALLOC(c->attn, B * T * d * e, false);
  CALLOC(c->ffnh, B * T * F * e, false);
  CALLOC(c->enc_out, B * T * d * e, false);
  // ---- decoder state ----
  const size_t Ld = c->Ld;
  // decoder K / V^T caches: fragment-major per (stream, head), key count padded to 64 (tw_kf_index / tw_vtf_index);
  // the padding is never written and must stay zero (V^T padding meets probability 0 in the P.V product)
  const size_t Pp = (P + 63) / 64 * 64;
  CALLOC(c->self_k, Ld * B * Pp * d * e, true);
  CALLOC(c->self_v, Ld * B * Pp * d * e, true);
  // cross K / V^T: bf16 (f32) fragment-major, or - MXFP8 contexts - e4m3 with one scale byte per (key, head) (tw_common.h)
  const size_t ckv_e = c->w8 ? 1 : e;
  CALLOC(c->cross_k, Ld * B * Tp * d * ckv_e, true);
  CALLOC(c->cross_v, Ld * B * Tp * d * ckv_e, true);
  if (c->w8) {
    CALLOC(c->cross_ksc, Ld * B * H * Tp, true);
    CALLOC(c->cross_vsc, Ld * B * H * Tp, true);
  }
  // per-token decoder activations that feed a projection are fragment-major in groups of 16 streams (tw_xt_index)
  // (sized for the 64 rows of a prefill launch - rows mode, tw_common.h - whatever max_batch is: < 1 MB)
  c->row_cap = (int)std::max<size_t>(B, 64);
  const size_t R = c->row_cap;
  const size_t Bg = (R + 15) / 16 * 16;  // whole groups of 16 streams
  CALLOC(c->dx0, Bg * d * e, true); CALLOC(c->dx1, Bg * d * e, true); CALLOC(c->dq, R * d * e, true);
  CALLOC(c->datt, Bg * d * e, true); CALLOC(c->dh, Bg * F * e, true);
  CALLOC(c->du, R * d * 4, true);
  CALLOC(c->dstats, R * (d / 4) * 2 * 4, true);
  c->row_ids_cap = (size_t)B * P;
  CALLOC(c->row_ids, c->row_ids_cap * 4, true);
  CHIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_rows), sizeof(int) * 2 * c->row_ids_cap, hipHostMallocDefault));
  CALLOC(c->row_lastts, c->row_ids_cap * 4, true);
  CALLOC(c->row_choice, c->row_ids_cap * 4, true);
  CALLOC(c->verify_state, (2 + 64) * 4, true);
  CHIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_verify), sizeof(int) * (2 + 64 + 2), hipHostMallocDefault));
  CALLOC(c->logits, R * V * 4, true);   // (the rows of a verify launch - up to 64 - whatever max_batch is: 13 MB)
  const size_t Ha = c->Ha > 0 ? c->Ha : 1;
  CALLOC(c->align, B * Ha * P * T * 4, true);
  CALLOC(c->align_slot, Ld * H * 4, false);
  {
    std::vector<int> slots(Ld * H, -1);
    for (int j = 0; j < c->Ha; ++j) {
      const int l = cfg->align_heads[2 * j], h = cfg->align_heads[2 * j + 1];
      if (l < 0 || l >= (int)Ld || h < 0 || h >= (int)H) { fail(c, TW_EINVAL, "alignment head (%d,%d) out of range", l, h); return bail(TW_EINVAL); }
      slots[(size_t)l * H + h] = j;
    }
    CHIP(hipMemcpy(c->align_slot, slots.data(), slots.size() * 4, hipMemcpyHostToDevice));
  }
  CALLOC(c->seq, B * P * 4, true); CALLOC(c->cur_ids, B * 4, true); CALLOC(c->finished, B * 4, true);
  CALLOC(c->last_ts, B * 4, true); CALLOC(c->stt, sizeof(DecState), true);
  CALLOC(c->begin_suppress_dev, 64 * 4, true); CALLOC(c->suppress_dev, 1024 * 4, true);
  CALLOC(c->suppress_bits, ((V + 31) / 32 + 2048) * 4, true);
  CALLOC(c->sampler_partials, 64 * 32 * sizeof(SamplerPartial), true);
  // ---- dtw workspace ----
  CALLOC(c->zbuf, B * Ha * P * T * 4, false);
  CALLOC(c->mat, B * P * T * 4, false);
  CALLOC(c->trace, B * (P + 1) * (T + 1), false);
  CALLOC(c->out_ts, B * (P + 1) * 4, false);
  CALLOC(c->n_cols, B * 4, true);
#undef CALLOC
#undef CHIP
  *out = c;
  return TW_OK;
}

// ---------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------
int tw_load_weight(tw_ctx* c, const char* name_c, const void* src, int32_t sdt, int32_t ndim, const int64_t* shape,
                   void* stream) {
  if (!c || !name_c || !src || !shape) return fail(c, TW_EINVAL, "tw_load_weight: null argument");
  if (c->shares_weights) return fail(c, TW_ESTATE, "tw_load_weight: this context shares another context's weights (tw_create_sibling)");
  if (sdt != TW_F32 && sdt != TW_BF16 && sdt != TW_F16) return fail(c, TW_EINVAL, "unsupported source dtype %d", sdt);
  TW_ON_DEVICE(c);
  hipStream_t st = pick_stream(c, stream);
  const std::string name(name_c);
  const int d = c->d, F = c->ffn;
  long long numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= shape[i];
  auto want = [&](std::initializer_list<long long> s) -> bool {
    if ((int)s.size() != ndim) return false;
    int i = 0;
    for (long long v : s) if (shape[i++] != v) return false;
    return true;
  };
  auto conv = [&](void* dst, float scale) -> int {
    HIPCHK(c, launch_convert(c->dtype, sdt, src, dst, numel, scale, st));
    c->loaded.insert(name);
    return TW_OK;
  };
  auto bad_shape = [&]() { return fail(c, TW_ENAME, "shape mismatch for %s", name_c); };

  if (name == "proj_out.weight") return TW_OK;  // tied to embed_tokens (HF:models/whisper/modeling_whisper.py:965)
  if (name == "model.encoder.conv1.weight") {
    if (!want({d, c->n_mels, 3})) return bad_shape();
    HIPCHK(c, launch_conv_weight_reorder(c->dtype, sdt, src, c->conv1_w, d, c->n_mels, c->C, st));
    c->loaded.insert(name);
    return TW_OK;
  }
  if (name == "model.encoder.conv2.weight") {
    if (!want({d, d, 3})) return bad_shape();
    HIPCHK(c, launch_conv_weight_reorder(c->dtype, sdt, src, c->conv2_w, d, d, d, st));
    c->loaded.insert(name);
    return TW_OK;
  }
  if (name == "model.encoder.conv1.bias") return want({d}) ? conv(c->conv1_b, 1.f) : bad_shape();
  if (name == "model.encoder.conv2.bias") return want({d}) ? conv(c->conv2_b, 1.f) : bad_shape();
  if (name == "model.encoder.embed_positions.weight") {
    if (ndim != 2 || shape[1] != d || shape[0] > 1500) return bad_shape();
    HIPCHK(c, launch_convert(TW_F32, sdt, src, c->enc_pos_raw, numel, 1.f, st));
    c->loaded.insert(name);
    c->enc_pos_rows = (int)shape[0];
    return TW_OK;
  }
  if (name == "model.encoder.layer_norm.weight") return want({d}) ? conv(c->enc_ln_g, 1.f) : bad_shape();
  if (name == "model.encoder.layer_norm.bias") return want({d}) ? conv(c->enc_ln_b, 1.f) : bad_shape();
  if (name == "model.decoder.embed_tokens.weight") return want({c->V, d}) ? conv(c->tok_emb, 1.f) : bad_shape();
  if (name == "model.decoder.embed_positions.weight") {
    if (ndim != 2 || shape[1] != d || shape[0] < c->P) return bad_shape();
    numel = (long long)c->P * d;
    return conv(c->dec_pos, 1.f);
  }
  if (name == "model.decoder.layer_norm.weight") return want({d}) ? conv(c->dec_ln_g, 1.f) : bad_shape();
  if (name == "model.decoder.layer_norm.bias") return want({d}) ? conv(c->dec_ln_b, 1.f) : bad_shape();

  // per-layer tensors: model.{encoder|decoder}.layers.{i}.<rest>
  bool is_dec = false;
  size_t pos = std::string::npos;
  const std::string pe = "model.encoder.layers.", pd = "model.decoder.layers.";
  if (name.compare(0, pe.size(), pe) == 0) pos = pe.size();
  else if (name.compare(0, pd.size(), pd) == 0) { pos = pd.size(); is_dec = true; }
  if (pos == std::string::npos) return fail(c, TW_ENAME, "unknown weight name %s", name_c);
  const size_t dot = name.find('.', pos);
  if (dot == std::string::npos) return fail(c, TW_ENAME, "unknown weight name %s", name_c);
  const int li = std::atoi(name.substr(pos, dot - pos).c_str());
  const std::string rest = name.substr(dot + 1);
  std::vector<LayerW>& Ls = is_dec ? c->dec : c->enc;
  if (li < 0 || li >= (int)Ls.size()) return fail(c, TW_ENAME, "layer index out of range in %s", name_c);
  LayerW& L = Ls[li];
  const size_t e = c->esz;
  const float qs = 0.125f;  // head_dim^-0.5 for head_dim 64, exact in bf16: folded into q_proj (HF scales q before QK^T, :309)
  const bool isw = want({d, d}), isb = want({d});
  if (rest == "self_attn_layer_norm.weight") return isb ? conv(L.ln1_g, 1.f) : bad_shape();
  if (rest == "self_attn_layer_norm.bias") return isb ? conv(L.ln1_b, 1.f) : bad_shape();
  if (rest == "final_layer_norm.weight") return isb ? conv(L.ln2_g, 1.f) : bad_shape();
  if (rest == "final_layer_norm.bias") return isb ? conv(L.ln2_b, 1.f) : bad_shape();
  if (rest == "self_attn.q_proj.weight") return isw ? conv(L.wqkv, qs) : bad_shape();
  if (rest == "self_attn.q_proj.bias") return isb ? conv(L.bqkv, qs) : bad_shape();
  if (rest == "self_attn.k_proj.weight") return isw ? conv(at(L.wqkv, (size_t)d * d, e), 1.f) : bad_shape();
  if (rest == "self_attn.v_proj.weight") return isw ? conv(at(L.wqkv, (size_t)2 * d * d, e), 1.f) : bad_shape();
  if (rest == "self_attn.v_proj.bias") return isb ? conv(at(L.bqkv, (size_t)2 * d, e), 1.f) : bad_shape();
  if (rest == "self_attn.out_proj.weight") return isw ? conv(L.wo, 1.f) : bad_shape();
  if (rest == "self_attn.out_proj.bias") return isb ? conv(L.bo, 1.f) : bad_shape();
  if (rest == "fc1.weight") return want({F, d}) ? conv(L.w1, 1.f) : bad_shape();
  if (rest == "fc1.bias") return want({F}) ? conv(L.b1, 1.f) : bad_shape();
  if (rest == "fc2.weight") return want({d, F}) ? conv(L.w2, 1.f) : bad_shape();
  if (rest == "fc2.bias") return isb ? conv(L.b2, 1.f) : bad_shape();
  if (is_dec) {
    if (rest == "encoder_attn_layer_norm.weight") return isb ? conv(L.lnx_g, 1.f) : bad_shape();
    if (rest == "encoder_attn_layer_norm.bias") return isb ? conv(L.lnx_b, 1.f) : bad_shape();
    if (rest == "encoder_attn.q_proj.weight") return isw ? conv(L.wq_c, qs) : bad_shape();
    if (rest == "encoder_attn.q_proj.bias") return isb ? conv(L.bq_c, qs) : bad_shape();
    if (rest == "encoder_attn.k_proj.weight") return isw ? conv(L.wkv_c, 1.f) : bad_shape();
    if (rest == "encoder_attn.v_proj.weight") return isw ? conv(at(L.wkv_c, (size_t)d * d, e), 1.f) : bad_shape();
    if (rest == "encoder_attn.v_proj.bias") return isb ? conv(at(L.bkv_c, (size_t)d, e), 1.f) : bad_shape();
    if (rest == "encoder_attn.out_proj.weight") return isw ? conv(L.wo_c, 1.f) : bad_shape();
    if (rest == "encoder_attn.out_proj.bias") return isb ? conv(L.bo_c, 1.f) : bad_shape();
  }
  return fail(c, TW_ENAME, "unknown weight name %s", name_c);
}

int tw_finalize_weights(tw_ctx* c, void* stream) {
  if (!c) return TW_EINVAL;
  if (c->shares_weights) return fail(c, TW_ESTATE, "tw_finalize_weights: this context shares another context's weights (tw_create_sibling)");
  TW_ON_DEVICE(c);
  hipStream_t st = pick_stream(c, stream);
  const size_t expect = 4 + 1 + 2 + (size_t)c->Le * 15 + 2 + 2 + (size_t)c->Ld * 24;
  if (c->loaded.size() != expect) {
    return fail(c, TW_ESTATE, "tw_finalize_weights: %zu of %zu tensors loaded", c->loaded.size(), expect);
  }
  const int n_old = c->enc_pos_rows;  // rows of the positional table as loaded
  if (c->T > n_old) return fail(c, TW_EINVAL, "source_positions %d exceeds the positional table (%d rows)", c->T, n_old);
  // A0: patch_hf_model (R:thestage_speechkit/nvidia/asr_pipeline.py:15-27)
  HIPCHK(c, launch_interp_positions(c->dtype, TW_F32, c->enc_pos_raw, c->enc_pos, n_old, c->T, c->d, st));
  if (c->finalized) return fail(c, TW_ESTATE, "tw_finalize_weights called twice");
  // fold every decoder pre-LayerNorm into the projection that consumes it (k_decode.hip): W <- W*g, gw, cb
  for (int l = 0; l < c->Ld; ++l) {
    LayerW& L = c->dec[l];
    HIPCHK(c, launch_fold_ln(c->dtype, L.wqkv, L.wqkv, L.ln1_g, L.ln1_b, L.bqkv, L.qkv_gw, L.qkv_cb, 3 * c->d, c->d, st));
    HIPCHK(c, launch_fold_ln(c->dtype, L.wq_c, L.wq_c, L.lnx_g, L.lnx_b, L.bq_c, L.qc_gw, L.qc_cb, c->d, c->d, st));
    HIPCHK(c, launch_fold_ln(c->dtype, L.w1, L.w1, L.ln2_g, L.ln2_b, L.b1, L.fc1_gw, L.fc1_cb, c->ffn, c->d, st));
    if (c->fuse_cq) {
      // cross query ahead.  With x1 = x0 + attn Wo^T + bo (self-attention block) and W' the folded cross-query weight,
      //   x1 W'^T = x0 W'^T + attn (W' Wo)^T + W' bo:
      // rows [3d,4d) of the QKV matrix = W' (same operand x0 as q|k|v; ln_cb carries c0 = W' bo, no LayerNorm of ITS input),
      // rows [d,2d) of the out-projection = W' Wo (same operand attn).  The LayerNorm of x1 is applied by the consumer.
      const size_t dd = (size_t)c->d * c->d;
      HIPCHK(c, hipMemcpyAsync(at(L.wqkv, 3 * dd, c->esz), L.wq_c, dd * c->esz, hipMemcpyDeviceToDevice, st));
      HIPCHK(c, launch_compose(c->dtype, L.wq_c, L.wo, L.bo, at(L.wo, dd, c->esz), L.qkv_cb + 3 * c->d, c->d, c->d, c->d, st));
    }
  }
  HIPCHK(c, launch_fold_ln(c->dtype, c->logit_w, c->tok_emb, c->dec_ln_g, c->dec_ln_b, nullptr, c->logit_gw, c->logit_cb,
                           c->V, c->d, st));
  // ... and the encoder's two per layer into its QKV / fc1 GEMMs (k_gemm.hip, GemmEpilogue::stats_in): same algebra, the row
  // statistics come from the GEMM that wrote the residual stream
  for (int l = 0; l < c->Le && c->enc_fold; ++l) {
    LayerW& L = c->enc[l];
    HIPCHK(c, launch_fold_ln(c->dtype, L.wqkv, L.wqkv, L.ln1_g, L.ln1_b, L.bqkv, L.qkv_gw, L.qkv_cb, 3 * c->d, c->d, st));
    HIPCHK(c, launch_fold_ln(c->dtype, L.w1, L.w1, L.ln2_g, L.ln2_b, L.b1, L.fc1_gw, L.fc1_cb, c->ffn, c->d, st));
  }
  // decoder projections are consumed by launch_gemv in the fragment-major layout (k_decode.hip: tile_weights_kernel)
  {
    const size_t e = c->esz;
    const size_t Vp = (size_t)(c->V + 15) / 16 * 16;
    size_t mx = Vp * c->d;
    if ((size_t)4 * c->d * c->d > mx) mx = (size_t)4 * c->d * c->d;
    if ((size_t)c->ffn * c->d > mx) mx = (size_t)c->ffn * c->d;
    if ((size_t)3 * c->C * c->d > mx) mx = (size_t)3 * c->C * c->d;
    void* scratch = nullptr;
    HIPCHK(c, hipMalloc(&scratch, mx * e));
    // narrow projections (N <= 2048: the three d x d matrices and fc2) are laid out for 8-row tiles so that their launches
    // cover 160 instead of 80 compute units (k_decode.hip); TW_SK_TR overrides (16 / 8 / 4) for experiments
    const char* tr_env = getenv("TW_SK_TR");
    const int tr_narrow = tr_env ? atoi(tr_env) : 8;
    auto retile = [&](void* w, int N, int K, unsigned char** scales, int* tr_out, bool narrow = false) -> int {
      const size_t Np = (size_t)(N + 15) / 16 * 16;
      int tr = 16;
      if (tr_out && (N <= 2048 || narrow) && (tr_narrow == 8 || (tr_narrow == 4 && !c->w8)) && N % 16 == 0) tr = tr_narrow;
      if (tr_out) *tr_out = tr;
      if (c->w8) {  // MXFP8 fragments + block scales replace the bf16 rows in the same buffer (half the bytes + 1/32)
        unsigned char* sc = reinterpret_cast<unsigned char*>(scratch);
        HIPCHK(c, launch_quant_mx8(w, sc, sc + Np * K, N, K, tr, st));
        HIPCHK(c, hipMemcpyAsync(w, scratch, Np * K + Np * K / 32, hipMemcpyDeviceToDevice, st));
        *scales = reinterpret_cast<unsigned char*>(w) + Np * K;
      } else {
        HIPCHK(c, launch_tile_weights(c->dtype, w, scratch, N, K, tr, st));
        HIPCHK(c, hipMemcpyAsync(w, scratch, Np * K * e, hipMemcpyDeviceToDevice, st));
        *scales = nullptr;
      }
      return TW_OK;
    };
    int r = TW_OK;
    for (int l = 0; l < c->Ld && r == TW_OK; ++l) {
      LayerW& L = c->dec[l];
      // (both layouts are tile-major, so the unfused launches simply read the leading 3d / d rows of the fused buffers)
      if ((r = retile(L.wqkv, (c->fuse_cq ? 4 : 3) * c->d, c->d, &L.s_qkv, nullptr)) != TW_OK) break;   // K/V scatter epilogue: 16-row tiles
      // (the fused out-projection of large-v3 has 2560 rows: 16-row tiles = 160 workgroups, measured 1.487 vs 1.510 ms per step
      //  with 320 workgroups of 8 rows, each of which re-reads the whole activation block)
      if ((r = retile(L.wo, (c->fuse_cq ? 2 : 1) * c->d, c->d, &L.s_o, &L.tr_o, getenv("TW_SK_TR_O8") != nullptr)) != TW_OK) break;
      if ((r = retile(L.wq_c, c->d, c->d, &L.s_qc, &L.tr_qc)) != TW_OK) break;
      if ((r = retile(L.wo_c, c->d, c->d, &L.s_oc, &L.tr_oc)) != TW_OK) break;
      if ((r = retile(L.w1, c->ffn, c->d, &L.s_1, &L.tr_1)) != TW_OK) break;
      if ((r = retile(L.w2, c->d, c->ffn, &L.s_2, &L.tr_2)) != TW_OK) break;
    }
    if (r == TW_OK) r = retile(c->logit_w, c->V, c->d, &c->logit_ws, nullptr);
    // the MFMA GEMMs (k_gemm.hip) read their weight operand fragment-major too (16-row tiles, never quantised): encoder
    // projections, the conv stem as GEMMs over [co][3*C] / [co][3*d] rows, and the decoder's cross K|V projection
    auto retile_gemm = [&](void* w, int N, int K) -> int {
      HIPCHK(c, launch_tile_weights(c->dtype, w, scratch, N, K, 16, st));
      HIPCHK(c, hipMemcpyAsync(w, scratch, (size_t)N * K * e, hipMemcpyDeviceToDevice, st));
      return TW_OK;
    };
    if (r == TW_OK) r = retile_gemm(c->conv1_w, c->d, 3 * c->C);
    if (r == TW_OK) r = retile_gemm(c->conv2_w, c->d, 3 * c->d);
    for (int l = 0; l < c->Le && r == TW_OK; ++l) {
      LayerW& L = c->enc[l];
      if ((r = retile_gemm(L.wqkv, 3 * c->d, c->d)) != TW_OK) break;
      if ((r = retile_gemm(L.wo, c->d, c->d)) != TW_OK) break;
      if ((r = retile_gemm(L.w1, c->ffn, c->d)) != TW_OK) break;
      if ((r = retile_gemm(L.w2, c->d, c->ffn)) != TW_OK) break;
    }
    for (int l = 0; l < c->Ld && r == TW_OK; ++l) r = retile_gemm(c->dec[l].wkv_c, 2 * c->d, c->d);
    hipError_t he = hipStreamSynchronize(st);
    hipFree(scratch);
    if (r != TW_OK) return r;
    HIPCHK(c, he);
  }
  c->finalized = true;
  return TW_OK;
}

// ---------------------------------------------------------------------------------------------
// A1 log-mel
// ---------------------------------------------------------------------------------------------
int tw_logmel(tw_ctx* c, const float* pcm, int64_t pcm_stride, const int32_t* n_valid_host, int32_t B, int32_t n_samples,
              void* out, int32_t out_dtype, void* stream) {
  if (!c || !pcm || !out) return fail(c, TW_EINVAL, "tw_logmel: null argument");
  TW_ON_DEVICE(c);
  if (B < 1 || B > c->Bmax) return fail(c, TW_EINVAL, "tw_logmel: B=%d outside [1,%d]", B, c->Bmax);
  if (n_samples < 400 || n_samples % 160 != 0) return fail(c, TW_EINVAL, "n_samples must be a multiple of 160 and >= 400");
  if ((size_t)B * c->n_mels * (n_samples / 160) > c->logspec_cap)
    return fail(c, TW_EINVAL, "tw_logmel: n_samples=%d exceeds the context capacity (%d frames)", n_samples, 2 * c->T);
  if (out_dtype != TW_F32 && out_dtype != c->dtype) return fail(c, TW_EINVAL, "out_dtype must be TW_F32 or the context dtype");
  hipStream_t st = pick_stream(c, stream);
  const int* nv = nullptr;
  if (n_valid_host) {
    memcpy(c->h_pinned + 512, n_valid_host, sizeof(int) * B);
    HIPCHK(c, hipMemcpyAsync(c->n_valid_dev, c->h_pinned + 512, sizeof(int) * B, hipMemcpyHostToDevice, st));
    nv = c->n_valid_dev;
  }
  tic(c, 0, st);
  HIPCHK(c, launch_logmel(pcm, pcm_stride, nv, B, n_samples, c->n_mels, c->lm, c->logspec_ws, c->lm_max, out, out_dtype, st));
  toc(c, 0, st);
  if (n_valid_host) HIPCHK(c, hipStreamSynchronize(st));  // pinned staging buffer is reused
  return TW_OK;
}

// ---------------------------------------------------------------------------------------------
// A2-A4 encoder
// ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

// encoder for B clips whose output goes to slots slot0 .. slot0+B-1 of enc_out (all other buffers are scratch from row 0)
int encode_core(tw_ctx* c, const void* mel, int32_t mel_dtype, int32_t B, int32_t slot0, void* out_hidden, int32_t out_dtype,
                void* stream) {
  if (!c || !mel) return fail(c, TW_EINVAL, "tw_encode: null argument");
  TW_ON_DEVICE(c);
  if (!c->finalized) return fail(c, TW_ESTATE, "tw_encode before tw_finalize_weights");
  if (B < 1 || slot0 < 0 || slot0 + B > c->Bmax) return fail(c, TW_EINVAL, "tw_encode: slots [%d, %d) outside [0,%d)", slot0, slot0 + B, c->Bmax);
  if (slot0 > c->encoded_B) return fail(c, TW_ESTATE, "tw_encode_at: slot0=%d but only %d slots are filled", slot0, c->encoded_B);
  if (mel_dtype != TW_F32 && mel_dtype != TW_BF16 && mel_dtype != TW_F16) return fail(c, TW_EINVAL, "tw_encode: mel_dtype %d is not TW_F32 / TW_BF16 / TW_F16", mel_dtype);
  if (out_hidden && out_dtype != TW_F32 && out_dtype != TW_BF16 && out_dtype != TW_F16)
    return fail(c, TW_EINVAL, "tw_encode: out_dtype %d is not TW_F32 / TW_BF16 / TW_F16", out_dtype);
  hipStream_t st = pick_stream(c, stream);
  const int d = c->d, T = c->T, C = c->C, H = c->H, F = c->ffn, dt = c->dtype;
  const size_t e = c->esz;
  const bool fold = c->enc_fold;
  tic(c, 1, st);
  HIPCHK(c, launch_mel_transpose(dt, mel_dtype, mel, c->melT, B, c->n_mels, 2 * T, C, st));
  {  // conv1 (k3,p1) + GELU as a GEMM over overlapping 3-row windows of the padded token-major mel
    GemmEpilogue ep{};
    ep.bias = c->conv1_b; ep.gelu = 1; ep.mode = EPI_ROWMAJOR;
    ep.c_map = RowMap{2 * T, (long long)(2 * T + 2) * d, d};
    ep.out = at(c->h1, d, e);
    HIPCHK(c, launch_gemm(dt, c->melT, RowMap{2 * T, (long long)(2 * T + 2) * C, C}, c->conv1_w, B * 2 * T, d, 3 * C, ep, st));
  }
  {  // conv2 (k3,s2,p1) + GELU + positions: windows of 3 rows at stride 2 rows
    GemmEpilogue ep{};
    ep.bias = c->conv2_b; ep.gelu = 1; ep.mode = EPI_ROWMAJOR;
    ep.res = c->enc_pos; ep.res_map = plain_rows(d); ep.res_mod = T;
    ep.c_map = plain_rows(d);
    ep.out = c->xa;
    if (fold) ep.stats_out = c->ln_stats_a;
    HIPCHK(c, launch_gemm(dt, c->h1, RowMap{T, (long long)(2 * T + 2) * d, 2LL * d}, c->conv2_w, B * T, d, 3 * d, ep, st));
  }
  const int M = B * T;
  for (int l = 0; l < c->Le; ++l) {
    const LayerW& L = c->enc[l];
    // pre-LayerNorms: folded into the consuming GEMM (weights carry the gain, the epilogue applies mean / rstd from the partial row
    // statistics the producing GEMM left), or - TW_ENC_FOLD_LN=0 - a launch that writes a normalised copy
    if (!fold) HIPCHK(c, launch_layernorm(dt, c->xa, L.ln1_g, L.ln1_b, c->lnbuf, M, d, st));
    {
      GemmEpilogue ep{};
      ep.bias = L.bqkv; ep.mode = EPI_QKV_ENC; ep.T = T; ep.Tp = c->Tp; ep.H = H;
      ep.out = c->qbuf; ep.out2 = c->kbuf; ep.out3 = c->vtbuf;
      if (fold) { ep.bias = nullptr; ep.stats_in = c->ln_stats_a; ep.stats_in_parts = d / 32; ep.ln_gw = L.qkv_gw; ep.ln_cb = L.qkv_cb; }
      HIPCHK(c, launch_gemm(dt, fold ? c->xa : c->lnbuf, plain_rows(d), L.wqkv, M, 3 * d, d, ep, st));
    }
    HIPCHK(c, launch_enc_attention(dt, c->qbuf, c->kbuf, c->vtbuf, c->attn, B, H, T, c->Tp, st));
    {
      GemmEpilogue ep{};
      ep.bias = L.bo; ep.mode = EPI_ROWMAJOR; ep.res = c->xa; ep.res_map = plain_rows(d); ep.c_map = plain_rows(d);
      ep.out = c->xb;
      if (fold) ep.stats_out = c->ln_stats_b;
      HIPCHK(c, launch_gemm(dt, c->attn, plain_rows(d), L.wo, M, d, d, ep, st));
    }
    if (!fold) HIPCHK(c, launch_layernorm(dt, c->xb, L.ln2_g, L.ln2_b, c->lnbuf, M, d, st));
    {
      GemmEpilogue ep{};
      ep.bias = L.b1; ep.gelu = 1; ep.mode = EPI_ROWMAJOR; ep.c_map = plain_rows(F); ep.out = c->ffnh;
      if (fold) { ep.bias = nullptr; ep.stats_in = c->ln_stats_b; ep.stats_in_parts = d / 32; ep.ln_gw = L.fc1_gw; ep.ln_cb = L.fc1_cb; }
      HIPCHK(c, launch_gemm(dt, fold ? c->xb : c->lnbuf, plain_rows(d), L.w1, M, F, d, ep, st));
    }
    {
      GemmEpilogue ep{};
      ep.bias = L.b2; ep.mode = EPI_ROWMAJOR; ep.res = c->xb; ep.res_map = plain_rows(d); ep.c_map = plain_rows(d);
      ep.out = c->xa;
      if (fold && l + 1 < c->Le) ep.stats_out = c->ln_stats_a;
      HIPCHK(c, launch_gemm(dt, c->ffnh, plain_rows(F), L.w2, M, d, F, ep, st));
    }
  }
  void* enc_dst = at(c->enc_out, (size_t)slot0 * T * d, e);
  HIPCHK(c, launch_layernorm(dt, c->xa, c->enc_ln_g, c->enc_ln_b, enc_dst, M, d, st));
  toc(c, 1, st);
  if (out_hidden) HIPCHK(c, launch_convert(out_dtype, dt, enc_dst, out_hidden, (long long)M * d, 1.f, st));
  c->encoded_B = slot0 + B;
  c->cross_B = slot0 < c->cross_B ? slot0 : c->cross_B;   // cross K/V of the re-encoded slots (and everything after) is stale
  return TW_OK;
}

// cross K/V of slots slot0 .. slot0+B-1: the per-(stream, head) blocks of a layer are contiguous per stream, so a slot offset is
// a pointer offset on the GEMM's input rows and on its head-split outputs
int cross_kv_core(tw_ctx* c, int32_t B, int32_t slot0, void* stream) {
  if (!c) return TW_EINVAL;
  TW_ON_DEVICE(c);
  if (B < 1 || slot0 < 0 || slot0 + B > c->encoded_B)
    return fail(c, TW_ESTATE, "tw_cross_kv: slots [%d, %d) but %d clips encoded", slot0, slot0 + B, c->encoded_B);
  if (slot0 > c->cross_B) return fail(c, TW_ESTATE, "tw_cross_kv_at: slot0=%d but cross K/V holds %d clips", slot0, c->cross_B);
  hipStream_t st = pick_stream(c, stream);
  const int d = c->d, T = c->T;
  const size_t per_layer = (size_t)c->Bmax * c->Tp * d;
  const size_t slot_off = (size_t)slot0 * c->Tp * d;             // elements of one layer's K (or V^T) arena before slot0
  const size_t sc_off = (size_t)slot0 * c->H * c->Tp;            // fp8 contexts: scale bytes before slot0
  tic(c, 2, st);
  for (int l = 0; l < c->Ld; ++l) {
    const LayerW& L = c->dec[l];
    GemmEpilogue ep{};
    ep.bias = L.bkv_c; ep.mode = c->w8 ? EPI_KV_CROSS8 : EPI_KV_CROSS; ep.T = T; ep.Tp = c->Tp; ep.H = c->H;
    ep.out = at(c->cross_k, per_layer * l + slot_off, c->w8 ? 1 : c->esz);
    ep.out2 = at(c->cross_v, per_layer * l + slot_off, c->w8 ? 1 : c->esz);
    if (c->w8) {
      ep.out3 = c->cross_ksc + (size_t)c->Bmax * c->H * c->Tp * l + sc_off;
      ep.out4 = c->cross_vsc + (size_t)c->Bmax * c->H * c->Tp * l + sc_off;
    }
    HIPCHK(c, launch_gemm(c->dtype, at(c->enc_out, (size_t)slot0 * T * d, c->esz), plain_rows(d), L.wkv_c, B * T, 2 * d, d, ep, st));
  }
  toc(c, 2, st);
  c->cross_B = slot0 + B;
  return TW_OK;
}

}  // namespace

extern "C" {

int tw_encode(tw_ctx* c, const void* mel, int32_t mel_dtype, int32_t B, void* out_hidden, int32_t out_dtype, void* stream) {
  if (c) { c->encoded_B = 0; c->cross_B = 0; }
  return encode_core(c, mel, mel_dtype, B, 0, out_hidden, out_dtype, stream);
}

int tw_encode_at(tw_ctx* c, const void* mel, int32_t mel_dtype, int32_t B, int32_t slot0, void* stream) {
  if (c && slot0 == 0) { c->encoded_B = 0; c->cross_B = 0; }
  return encode_core(c, mel, mel_dtype, B, slot0, nullptr, 0, stream);
}

int tw_cross_kv(tw_ctx* c, int32_t B, void* stream) {
  if (c) c->cross_B = 0;
  return cross_kv_core(c, B, 0, stream);
}

int tw_cross_kv_at(tw_ctx* c, int32_t B, int32_t slot0, void* stream) { return cross_kv_core(c, B, slot0, stream); }

int tw_adopt_cross_kv(tw_ctx* dst, int32_t dst_slot0, tw_ctx* src, int32_t src_slot0, int32_t B, void* stream, void* src_stream) {
  if (!dst || !src) return fail(dst, TW_EINVAL, "tw_adopt_cross_kv: null context");
  TW_ON_DEVICE(dst);
  if (dst == src) return fail(dst, TW_EINVAL, "tw_adopt_cross_kv: source and destination are the same context");
  if (dst->cfg.device != src->cfg.device || dst->d != src->d || dst->H != src->H || dst->Ld != src->Ld || dst->T != src->T ||
      dst->Tp != src->Tp || dst->dtype != src->dtype || dst->w8 != src->w8)
    return fail(dst, TW_EINVAL, "tw_adopt_cross_kv: the contexts differ (model dimensions, source_positions, dtype or device)");
  if (B < 1 || src_slot0 < 0 || src_slot0 + B > src->cross_B)
    return fail(dst, TW_ESTATE, "tw_adopt_cross_kv: source slots [%d, %d) but the source holds cross K/V of %d clips", src_slot0, src_slot0 + B, src->cross_B);
  if (dst_slot0 < 0 || dst_slot0 + B > dst->Bmax) return fail(dst, TW_EINVAL, "tw_adopt_cross_kv: destination slots [%d, %d) outside [0,%d)", dst_slot0, dst_slot0 + B, dst->Bmax);
  if (dst_slot0 > dst->cross_B) return fail(dst, TW_ESTATE, "tw_adopt_cross_kv: dst_slot0=%d but the destination holds %d clips", dst_slot0, dst->cross_B);
  hipStream_t st = pick_stream(dst, stream);
  hipStream_t sst = pick_stream(src, src_stream);
  // the copies follow everything `src` has enqueued so far (its encoder + cross-K/V launches) ...
  if (sst != st) {
    HIPCHK(dst, hipEventRecord(dst->ring_ev[0], sst));
    HIPCHK(dst, hipStreamWaitEvent(st, dst->ring_ev[0], 0));
  }
  // one 2-D copy per arena: row l = layer l, `width` bytes of B consecutive slots, pitches = one layer of each context
  const size_t slot_bytes = (size_t)dst->Tp * dst->d * (dst->w8 ? 1 : dst->esz);
  auto copy = [&](void* d_base, const void* s_base, size_t per_slot) -> hipError_t {
    return hipMemcpy2DAsync(reinterpret_cast<char*>(d_base) + (size_t)dst_slot0 * per_slot, (size_t)dst->Bmax * per_slot,
                            reinterpret_cast<const char*>(s_base) + (size_t)src_slot0 * per_slot, (size_t)src->Bmax * per_slot,
                            (size_t)B * per_slot, (size_t)dst->Ld, hipMemcpyDeviceToDevice, st);
  };
  HIPCHK(dst, copy(dst->cross_k, src->cross_k, slot_bytes));
  HIPCHK(dst, copy(dst->cross_v, src->cross_v, slot_bytes));
  if (dst->w8) {
    const size_t sc = (size_t)dst->H * dst->Tp;
    HIPCHK(dst, copy(dst->cross_ksc, src->cross_ksc, sc));
    HIPCHK(dst, copy(dst->cross_vsc, src->cross_vsc, sc));
  }
  // ... and whatever `src` enqueues next (it may refill these slots) follows the copies
  if (sst != st) {
    HIPCHK(dst, hipEventRecord(dst->ring_ev[1], st));
    HIPCHK(dst, hipStreamWaitEvent(sst, dst->ring_ev[1], 0));
  }
  dst->cross_B = dst_slot0 + B;
  if (dst->encoded_B < dst->cross_B) dst->encoded_B = dst->cross_B;   // (the encoder states themselves stay in `src`: nothing reads them after A5)
  return TW_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------------------
namespace {

// TW_FUSE_CQ=0 at run time: the unfused launch sequence on the same (fused-layout) weights, for A/B measurements
bool getenv_off_fuse() {
  static const bool off = []() { const char* e = getenv("TW_FUSE_CQ"); return e && atoi(e) == 0; }();
  return off;
}

// one token for every stream: embed -> Ld layers -> final LN + tied logits (fp32)
// rs > 0: rows mode (tw_common.h: tw_row_of) - the B rows are rs streams x B / rs consecutive positions whose tokens are `ids`
// (device, row order); no logits are produced (the tokens of those positions are known: prefill_core)
// embed = false: the input rows are already in place (the sampler's last launch of the previous step wrote them, SamplerArgs::x_next)
// rows_logits: rows mode WITH the tied-logits projection of every row (draft-and-verify: verify_round) - the same launch the step uses,
// so a row's logits are bit for bit what the one-position step would have produced (k_decode.hip: one reduction order for every B)
int decode_core(tw_ctx* c, int B, hipStream_t st, int rs = 0, const int* ids = nullptr, bool embed = true, bool rows_logits = false) {
  const int d = c->d, H = c->H, F = c->ffn, T = c->T, P = c->P, dt = c->dtype;
  const size_t e = c->esz;
  if (embed) HIPCHK(c, launch_embed(dt, rs > 0 ? ids : c->cur_ids, c->stt, c->tok_emb, c->dec_pos, c->dx0, B, d, rs, st));
  void* xin = c->dx0;
  void* xmid = c->dx1;
  const int Pp = (P + 63) / 64 * 64;
  const size_t self_layer = (size_t)c->Bmax * Pp * d;
  const size_t cross_layer = (size_t)c->Bmax * c->Tp * d;
  for (int l = 0; l < c->Ld; ++l) {
    const LayerW& L = c->dec[l];
    void* sk = at(c->self_k, self_layer * l, e);
    void* sv = at(c->self_v, self_layer * l, e);
    // "cross query ahead" (tw_ctx::fuse_cq): the cross-attention query u = x1 . W'^T is accumulated by the two projections
    // before it - x0 . W'^T as a fourth segment of the QKV launch, attn . (W' Wo)^T as a second half of the out-projection
    // launch - and the cross attention applies the folded LayerNorm of x1 itself: 7 instead of 8 dependent launches per layer
    const bool fq_on = c->fuse_cq && !getenv_off_fuse() && d / (L.tr_o ? L.tr_o : 16) <= 192;
    {  // LN + fused QKV; k,v rows go straight into the cache at position pos
      GemvArgs a{};
      a.x = xin; a.ldx = d; a.ln_gw = L.qkv_gw; a.ln_cb = L.qkv_cb; a.W = L.wqkv; a.wscale = L.s_qkv; a.a16 = c->a16; a.N = (fq_on ? 4 : 3) * d; a.K = d; a.B = B;
      a.y = c->dq; a.ldy = d; a.kcache = sk; a.vcache = sv; a.cache_bstride = (long long)Pp * d; a.cache_hstride = (long long)Pp * 64; a.d_model = d; a.stt = c->stt;
      a.u = fq_on ? c->du : nullptr;
      a.rows_streams = rs;
      HIPCHK(c, launch_gemv(dt, a, st));
    }
    HIPCHK(c, launch_dec_self_attn(dt, c->dq, sk, sv, Pp, c->datt, B, H, c->dec_key_bound > 0 ? c->dec_key_bound : P, c->stt, rs, st));
    {
      GemvArgs a{};
      a.x = c->datt; a.ldx = d; a.W = L.wo; a.wscale = L.s_o; a.a16 = c->a16; a.tr = L.tr_o; a.bias = L.bo; a.N = (fq_on ? 2 : 1) * d; a.K = d; a.B = B;
      a.res = xin; a.ldres = d; a.y = xmid; a.ldy = d;
      if (fq_on) { a.u = c->du; a.nsplit = d; a.stats = c->dstats; }
      HIPCHK(c, launch_gemv(dt, a, st));
    }
    FusedQ fq{};
    if (fq_on) {
      fq.u = c->du; fq.stats = c->dstats; fq.n_part = d / (L.tr_o ? L.tr_o : 16); fq.gw = L.qc_gw; fq.cb = L.qc_cb; fq.d = d;
    } else {
      GemvArgs a{};
      a.x = xmid; a.ldx = d; a.ln_gw = L.qc_gw; a.ln_cb = L.qc_cb; a.W = L.wq_c; a.wscale = L.s_qc; a.a16 = c->a16; a.tr = L.tr_qc; a.N = d; a.K = d; a.B = B;
      a.y = c->dq; a.ldy = d;
      HIPCHK(c, launch_gemv(dt, a, st));
    }
    HIPCHK(c, launch_dec_cross_attn(dt, c->dq, fq, at(c->cross_k, cross_layer * l, c->w8 ? 1 : e), at(c->cross_v, cross_layer * l, c->w8 ? 1 : e),
                                    c->datt, B, H, T, c->Tp, c->Ha > 0 ? c->align_slot + (size_t)l * H : nullptr, c->align, c->Ha,
                                    P, c->stt, c->w8 ? c->cross_ksc + (size_t)c->Bmax * H * c->Tp * l : nullptr,
                                    c->w8 ? c->cross_vsc + (size_t)c->Bmax * H * c->Tp * l : nullptr, rs, st));
    {
      GemvArgs a{};
      a.x = c->datt; a.ldx = d; a.W = L.wo_c; a.wscale = L.s_oc; a.a16 = c->a16; a.tr = L.tr_oc; a.bias = L.bo_c; a.N = d; a.K = d; a.B = B; a.res = xmid; a.ldres = d;
      a.y = xin; a.ldy = d;
      HIPCHK(c, launch_gemv(dt, a, st));
    }
    {
      GemvArgs a{};
      a.x = xin; a.ldx = d; a.ln_gw = L.fc1_gw; a.ln_cb = L.fc1_cb; a.W = L.w1; a.wscale = L.s_1; a.a16 = c->a16; a.tr = L.tr_1; a.N = F; a.K = d; a.B = B;
      a.gelu = 1; a.y = c->dh; a.ldy = F;
      HIPCHK(c, launch_gemv(dt, a, st));
    }
    {
      GemvArgs a{};
      a.x = c->dh; a.ldx = F; a.W = L.w2; a.wscale = L.s_2; a.a16 = c->a16; a.tr = L.tr_2; a.bias = L.b2; a.N = d; a.K = F; a.B = B; a.res = xin; a.ldres = d;
      a.y = xmid; a.ldy = d;
      HIPCHK(c, launch_gemv(dt, a, st));
    }
    void* t = xin; xin = xmid; xmid = t;
  }
  if (rs == 0 || rows_logits) {
    GemvArgs a{};
    a.x = xin; a.ldx = d; a.ln_gw = c->logit_gw; a.ln_cb = c->logit_cb; a.W = c->logit_w; a.wscale = c->logit_ws; a.a16 = c->a16; a.N = c->V; a.K = d; a.B = B;
    a.y_f32 = c->logits;
    HIPCHK(c, launch_gemv(dt, a, st));
  }
  return TW_OK;
}

// Batched prefill of the positions [0, n_pos) of B streams whose tokens are known (host table tok[b * ld + p]): launches of up to
// `cap` rows = B streams x L consecutive positions (rows mode), each a full pass through the decoder layers that fills the
// self-attention caches and the alignment rows of its positions; the weights are streamed once per L positions instead of once per
// position.  On return the device position is n_pos.  Not captured in a graph (a handful of launches per call).
int prefill_core(tw_ctx* c, int B, const int* tok, int ld, int n_pos, hipStream_t st) {
  if (n_pos <= 0) return TW_OK;
  const int cap = (c->w8 && !c->a16) ? 16 : 64;     // W8A8 quantises activations per group of 16 rows: one group per launch
  if (B > cap) return fail(c, TW_EINVAL, "prefill: %d streams exceed the %d rows of a launch", B, cap);
  const int L = std::max(1, std::min(cap, c->row_cap) / B);
  // position-major token table: row r of the launch at p0 is stream r % B at position p0 + r / B
  if ((size_t)n_pos * B > c->row_ids_cap) return fail(c, TW_EINVAL, "prefill: %d positions x %d streams exceed the row table", n_pos, B);
  int* h = c->h_rows;
  for (int p = 0; p < n_pos; ++p)
    for (int b = 0; b < B; ++b) h[(size_t)p * B + b] = tok[(size_t)b * ld + p];
  HIPCHK(c, hipMemcpyAsync(c->row_ids, h, sizeof(int) * (size_t)n_pos * B, hipMemcpyHostToDevice, st));
  for (int p0 = 0; p0 < n_pos; p0 += L) {
    const int l = std::min(L, n_pos - p0);
    c->dec_key_bound = std::min(((p0 + l + 63) / 64) * 64, ((c->P + 63) / 64) * 64);
    int r = decode_core(c, l * B, st, B, c->row_ids + (size_t)p0 * B);
    if (r != TW_OK) return r;
    HIPCHK(c, launch_advance(c->stt, l, st));
  }
  return TW_OK;
}

// One ROUND of draft-and-verify (tw_greedy_opts::n_draft).  tok[b * ld + p] holds, for every stream, the true tokens at
// positions <= p_start and GUESSES behind them; the device position is p_start.  Positions [p_start, p_end] are run in rows mode
// (launches of up to rows_cap rows = B streams x consecutive positions, each with the logits of every row and the sampler's choice for
// every row - the processors see the given tokens as history), and the round ends at the first position p_acc where some stream's
// choice differs from the token the next row was given (or at p_end): the choices at p_acc become the tokens at p_acc + 1 - exactly
// what the one-position loop would have produced there, because a row's logits do not depend on which other rows share its launch
// (k_decode.hip) and every earlier given token was confirmed - and the device state (history, last timestamp, finished flags,
// position) is the loop's state after that token.  The host synchronises once per launch (it needs the decision to go on).
// On return *p_next = p_acc + 1 and tok[.. + *p_next] holds those tokens.
// *next_given = 1 when the tokens the round PRODUCED at *p_next are (for every stream) the guesses that stood there (p_next <= p_given).
int verify_round(tw_ctx* c, int B, int* tok, int ld, int n_begin, int ts_begin, int p_start, int p_end, int rows_cap, const SamplerArgs& sa0,
                 hipStream_t st, int* p_next, int* n_launches, int p_given, int* next_given) {
  const int n_pos = p_end - p_start + 1;
  if (n_pos < 1 || B > rows_cap) return fail(c, TW_EINVAL, "verify: bad round [%d, %d] for %d streams", p_start, p_end, B);
  if ((size_t)n_pos * B > c->row_ids_cap) return fail(c, TW_EINVAL, "verify: %d positions x %d streams exceed the row table", n_pos, B);
  const int L = std::max(1, rows_cap / B);
  int* h = c->h_rows;
  int* hl = c->h_rows + c->row_ids_cap;
  for (int b = 0; b < B; ++b) {
    int lt = -1;
    for (int p = n_begin; p < p_start; ++p) if (tok[(size_t)b * ld + p] >= ts_begin) lt = tok[(size_t)b * ld + p];
    for (int p = p_start; p <= p_end; ++p) {
      const int t = tok[(size_t)b * ld + p];
      if (p >= n_begin && t >= ts_begin) lt = t;
      h[(size_t)(p - p_start) * B + b] = t;
      hl[(size_t)(p - p_start) * B + b] = lt;
    }
  }
  int* hv = c->h_verify;
  hv[66] = 0x7fffffff; hv[67] = 0;
  HIPCHK(c, hipMemcpyAsync(c->row_ids, h, sizeof(int) * (size_t)n_pos * B, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->row_lastts, hl, sizeof(int) * (size_t)n_pos * B, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->verify_state, hv + 66, sizeof(int) * 2, hipMemcpyHostToDevice, st));
  for (int p0 = p_start; p0 <= p_end; p0 += L) {
    const int l = std::min(L, p_end - p0 + 1);
    const size_t off = (size_t)(p0 - p_start) * B;
    c->dec_key_bound = std::min(((p0 + l + 63) / 64) * 64, ((c->P + 63) / 64) * 64);
    int r = decode_core(c, l * B, st, B, c->row_ids + off, true, true);
    if (r != TW_OK) return r;
    SamplerArgs sa = sa0;
    sa.B = l * B; sa.rows_streams = B; sa.row_pos0 = p0; sa.row_lastts = c->row_lastts + off; sa.row_choice = c->row_choice + off;
    sa.row_last_pos = p_end; sa.row_is_last = (p0 + l > p_end) ? 1 : 0; sa.verify_state = c->verify_state;
    HIPCHK(c, launch_sampler_rows(sa, st));
    HIPCHK(c, hipMemcpyAsync(hv, c->verify_state, sizeof(int) * (2 + B), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    ++*n_launches;
    if (hv[1] != 0) {
      const int pn = hv[1];
      if (pn < p0 + 1 || pn > p0 + l) return fail(c, TW_EHIP, "verify: decided position %d outside the launch [%d, %d)", pn - 1, p0, p0 + l);
      bool same = pn <= p_given;
      for (int b = 0; b < B; ++b) {
        same = same && tok[(size_t)b * ld + pn] == hv[2 + b];
        tok[(size_t)b * ld + pn] = hv[2 + b];
      }
      *next_given = same ? 1 : 0;
      *p_next = pn;
      return TW_OK;
    }
  }
  return fail(c, TW_EHIP, "verify: the round ended without a decision");
}

int reset_state(tw_ctx* c, int n_prompt, hipStream_t st) {
  DecState s{0, n_prompt, 0, 0};
  memcpy(c->h_pinned + 576, &s, sizeof s);
  HIPCHK(c, hipMemcpyAsync(c->stt, c->h_pinned + 576, sizeof s, hipMemcpyHostToDevice, st));
  return TW_OK;
}

}  // namespace

extern "C" {

int tw_decoder_reset(tw_ctx* c, int32_t B, void* stream) {
  if (!c) return TW_EINVAL;
  TW_ON_DEVICE(c);
  if (B < 1 || B > c->cross_B) return fail(c, TW_ESTATE, "tw_decoder_reset: B=%d but cross K/V holds %d clips", B, c->cross_B);
  hipStream_t st = pick_stream(c, stream);
  int r = reset_state(c, 0, st);
  if (r != TW_OK) return r;
  c->dec_key_bound = c->P;
  HIPCHK(c, hipStreamSynchronize(st));
  return TW_OK;
}

int tw_decode_step(tw_ctx* c, int32_t B, const int32_t* ids_host, float* logits_dev, void* stream) {
  if (!c || !ids_host) return fail(c, TW_EINVAL, "tw_decode_step: null argument");
  TW_ON_DEVICE(c);
  if (B < 1 || B > c->cross_B) return fail(c, TW_ESTATE, "tw_decode_step: B=%d but cross K/V holds %d clips", B, c->cross_B);
  hipStream_t st = pick_stream(c, stream);
  HIPCHK(c, hipMemcpyAsync(c->cur_ids, ids_host, sizeof(int) * B, hipMemcpyHostToDevice, st));
  int r = decode_core(c, B, st);
  if (r != TW_OK) return r;
  if (logits_dev)
    HIPCHK(c, hipMemcpyAsync(logits_dev, c->logits, sizeof(float) * (size_t)B * c->V, hipMemcpyDeviceToDevice, st));
  HIPCHK(c, launch_advance(c->stt, 1, st));  // pos += 1 on the device (keeps the step graph-compatible)
  return TW_OK;
}

int tw_generate_greedy(tw_ctx* c, int32_t B, const int32_t* prompt, int32_t n_prompt, const tw_greedy_opts* o,
                       int32_t* out_ids, int32_t* out_len, void* stream) {
  if (!c || !prompt || !o || !out_ids || !out_len) return fail(c, TW_EINVAL, "tw_generate_greedy: null argument");
  TW_ON_DEVICE(c);
  if (B < 1 || B > c->cross_B) return fail(c, TW_ESTATE, "tw_generate_greedy: B=%d but cross K/V holds %d clips", B, c->cross_B);
  if (n_prompt < 1 || n_prompt >= c->P) return fail(c, TW_EINVAL, "bad n_prompt %d", n_prompt);
  if (o->n_begin_suppress > 64 || o->n_suppress > 1024) return fail(c, TW_EINVAL, "suppress lists too long");
  if (o->want_alignment && c->Ha == 0) return fail(c, TW_EINVAL, "want_alignment but the context has no alignment heads");
  // forced output tokens (tw_greedy_opts::n_forced): the begin index of the generation is n_begin, the tokens behind it are given
  const int n_forced = o->n_forced;
  if (n_forced < 0 || n_forced >= n_prompt) return fail(c, TW_EINVAL, "bad n_forced %d (n_prompt %d)", n_forced, n_prompt);
  // draft tokens (tw_greedy_opts::n_draft): guesses of the output, verified - the result is the plain call's, token for token
  const int n_draft = o->n_draft;
  if (n_draft < 0 || n_draft >= n_prompt || (n_draft > 0 && n_forced > 0))
    return fail(c, TW_EINVAL, "bad n_draft %d (n_prompt %d, n_forced %d: one of the two)", n_draft, n_prompt, n_forced);
  if (o->pad_id < 0 || o->pad_id >= c->V || o->eos_id < 0 || o->eos_id >= c->V)
    return fail(c, TW_EINVAL, "pad_id %d / eos_id %d outside the vocabulary", o->pad_id, o->eos_id);   // (the sampler embeds pad_id for finished rows)
  const int n_begin = n_prompt - n_forced - n_draft;
  int max_len = n_begin + o->max_new_tokens;
  if (o->max_length > 0 && o->max_length < max_len) max_len = o->max_length;
  if (max_len > c->P) max_len = c->P;
  if (max_len <= n_prompt) return fail(c, TW_EINVAL, "nothing to generate (max_len %d <= n_prompt %d)", max_len, n_prompt);
  const int out_ld = o->max_length > 0 ? o->max_length : max_len;
  c->dec_key_bound = max_len;   // per step: the 64-key bucket of the position (below)
  hipStream_t st = pick_stream(c, stream);
  const int P = c->P;

  // ---- initial state: staged in PINNED memory, so the uploads are asynchronous and the loop's first launch follows them without a
  //      host synchronisation (the staging area is not touched again before the stream synchronisation at the end of this call) ----
  int* hseq = c->h_stage;                                  // [B][P]
  int* first = hseq + (size_t)c->Bmax * P;                 // [B]
  int* zeros = first + c->Bmax;
  int* neg = zeros + c->Bmax;
  int* bsup = neg + c->Bmax;                               // [64]
  int* sup = bsup + 64;                                    // [1024]
  DecState* s0p = reinterpret_cast<DecState*>(sup + 1024);
  for (size_t i = 0; i < (size_t)B * P; ++i) hseq[i] = o->pad_id;
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < n_prompt; ++i) {
      const int t = prompt[(size_t)b * n_prompt + i];
      if (t < 0 || t >= c->V) return fail(c, TW_EINVAL, "prompt token %d out of range", t);
      hseq[(size_t)b * P + i] = t;
    }
    // with forced tokens the loop starts at the LAST given position (everything before it is prefilled below)
    first[b] = prompt[(size_t)b * n_prompt + (n_forced > 0 ? n_prompt - 1 : 0)];
    zeros[b] = 0;
    neg[b] = -1;
    for (int i = n_begin; i < n_prompt; ++i) {   // state the sampler would have after producing the forced tokens itself
      const int t = prompt[(size_t)b * n_prompt + i];
      if (t == o->eos_id) return fail(c, TW_EINVAL, n_draft > 0 ? "a draft token is <eos>" : "a forced token is <eos>");
      if (n_forced > 0 && o->timestamps && t > o->no_timestamps_id) neg[b] = t;
    }
  }
  HIPCHK(c, hipMemcpyAsync(c->seq, hseq, sizeof(int) * (size_t)B * P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->cur_ids, first, sizeof(int) * B, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->finished, zeros, sizeof(int) * B, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->last_ts, neg, sizeof(int) * B, hipMemcpyHostToDevice, st));
  if (o->n_begin_suppress > 0) {
    memcpy(bsup, o->begin_suppress, sizeof(int) * o->n_begin_suppress);
    HIPCHK(c, hipMemcpyAsync(c->begin_suppress_dev, bsup, sizeof(int) * o->n_begin_suppress, hipMemcpyHostToDevice, st));
  }
  if (o->n_suppress > 0) {
    memcpy(sup, o->suppress, sizeof(int) * o->n_suppress);
    HIPCHK(c, hipMemcpyAsync(c->suppress_dev, sup, sizeof(int) * o->n_suppress, hipMemcpyHostToDevice, st));
  }
  HIPCHK(c, launch_suppress_bitmap(c->suppress_dev, o->n_suppress, c->suppress_bits, c->V, st));
  *s0p = DecState{0, n_begin, max_len - 1, B};
  HIPCHK(c, hipMemcpyAsync(c->stt, s0p, sizeof(DecState), hipMemcpyHostToDevice, st));
  int s_start = n_forced > 0 ? n_prompt - 1 : 0;
  if (n_forced > 0) {   // positions 0 .. n_prompt-2 in batched launches (self-attention caches + alignment rows); device pos -> s_start
    int r = prefill_core(c, B, prompt, n_prompt, s_start, st);
    if (r != TW_OK) return r;
  }

  SamplerArgs sa{};
  sa.logits = c->logits; sa.V = c->V; sa.B = B; sa.seq = c->seq; sa.seq_ld = P; sa.cur_ids = c->cur_ids;
  sa.finished = c->finished; sa.last_ts = c->last_ts; sa.stt = c->stt;
  sa.eos = o->eos_id; sa.pad = o->pad_id; sa.min_new = o->min_new_tokens; sa.timestamps = o->timestamps;
  sa.no_ts_id = o->no_timestamps_id; sa.max_initial_ts = o->max_initial_timestamp_index;
  sa.begin_suppress = c->begin_suppress_dev; sa.n_begin_suppress = o->n_begin_suppress;
  sa.suppress_bits = c->suppress_bits; sa.partials = c->sampler_partials;
  // the embedding of the token a step has chosen is written by the sampler's last launch (k_decode.hip: embed_row), so a step is
  // layers + logits + sampler; only the FIRST step of the call needs its input row from a launch of its own (TW_FUSE_EMBED=0: A/B)
  static const bool fuse_embed = []() { const char* e = getenv("TW_FUSE_EMBED"); return !(e && atoi(e) == 0); }();
  c->last_draft[0] = n_draft * B; c->last_draft[1] = c->last_draft[2] = c->last_draft[3] = 0;
  bool draft_all_done = false;
  if (n_draft > 0) {
    // ---- draft-and-verify: rounds of rows-mode launches over the given tokens (verify_round), then the ordinary loop from where the
    //      last round stopped.  Round 1 takes everything that was offered (prompt included: its positions have to be processed anyway);
    //      after a rejection the REST of the draft is offered again at the same positions (a changed timestamp or word usually leaves
    //      the text behind it as it was), in launches of one group of rows - such a launch costs about what one step costs and
    //      yields at least the one token a step yields - until a round confirms nothing or the draft is used up. ----
    static const int retry_rows = []() { const char* e = getenv("TW_DRAFT_RETRY_ROWS"); const int v = e ? atoi(e) : 16; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    static const int max_rounds = []() { const char* e = getenv("TW_DRAFT_MAX_ROUNDS"); const int v = e ? atoi(e) : 16; return v < 1 ? 1 : v; }();
    static const int first_rows = []() { const char* e = getenv("TW_DRAFT_FIRST_ROWS"); const int v = e ? atoi(e) : 64; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    const int cap = std::min((c->w8 && !c->a16) ? 16 : 64, c->row_cap);   // W8A8 quantises activations per group of 16 rows: one group per launch
    if (B > cap) return fail(c, TW_EINVAL, "n_draft: %d streams exceed the %d rows of a launch", B, cap);
    const int ts_begin = o->timestamps ? o->no_timestamps_id + 1 : c->V;
    const int p_end = n_prompt - 1;
    tic(c, 3, st);
    int pos = 0, rounds = 0;
    while (true) {
      // (several streams: at least four positions per retry launch, or a round could not confirm anything)
      const int rows = rounds == 0 ? std::min(cap, std::max(first_rows, B)) : std::min(cap, std::max(retry_rows, 4 * B));
      const int pe = rounds == 0 ? p_end : std::min(p_end, pos + std::max(1, rows / B) - 1);
      int pn = 0, next_given = 0;
      int r = verify_round(c, B, hseq, P, n_begin, ts_begin, pos, pe, rows, sa, st, &pn, &c->last_draft[2], p_end, &next_given);
      if (r != TW_OK) return r;
      ++rounds;
      // guesses this round confirmed: the given tokens at positions <= p_acc that were compared, and the token the round produced
      // itself at p_acc + 1 where it is the guess that stood there (a round that ends without a mismatch before the draft does)
      const int confirmed = pn - 1 - std::max(pos, n_begin - 1) + ((next_given && pn >= n_begin) ? 1 : 0);
      c->last_draft[1] += std::max(0, confirmed) * B;
      pos = pn;
      draft_all_done = true;
      for (int b = 0; b < B; ++b) draft_all_done &= pos >= n_begin && hseq[(size_t)b * P + pos] == o->eos_id;
      if (draft_all_done || pos > p_end - 1 || pos >= max_len - 1 || rounds >= max_rounds || (rounds > 1 && confirmed <= 0)) break;
    }
    c->last_draft[3] = rounds;
    s_start = pos;
  }
  if (fuse_embed) {
    sa.tok_emb = c->tok_emb; sa.pos_emb = c->dec_pos; sa.x_next = c->dx0; sa.d = c->d; sa.dtype = c->dtype;
    HIPCHK(c, launch_embed(c->dtype, c->cur_ids, c->stt, c->tok_emb, c->dec_pos, c->dx0, B, c->d, 0, st));
  }

  // ---- graph replay: one captured step per self-attention length bucket, captured on first use ----
  char keybuf[256];
  snprintf(keybuf, sizeof keybuf, "%d|%d|%d|%d|%d|%d|%d|%d|%d|%d", B, o->eos_id, o->pad_id, o->min_new_tokens, o->timestamps,
           o->no_timestamps_id, o->max_initial_timestamp_index, o->n_begin_suppress, o->n_suppress, (int)fuse_embed);
  const bool use_graph = c->cfg.use_graph != 0;
  if (use_graph && c->step_graph_key != keybuf) {   // other options: the captured sampler arguments are stale
    for (auto& kv : c->step_graphs) (void)hipGraphExecDestroy(kv.second);
    c->step_graphs.clear();
    c->step_graph_key = keybuf;
  }
  // `n` consecutive steps per graph (the step is position-independent: the position lives in device memory): fewer graph
  // launches for the host to issue when a step is short (turbo, one stream: 0.2 ms)
  auto graph_for = [&](int kb, int n, hipGraphExec_t* out) -> int {
    const std::string k = std::to_string(kb) + "x" + std::to_string(n);
    auto it = c->step_graphs.find(k);
    if (it != c->step_graphs.end()) { *out = it->second; return TW_OK; }
    hipGraph_t g = nullptr;
    c->dec_key_bound = kb;
    HIPCHK(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int r = TW_OK;
    hipError_t es = hipSuccess;
    for (int i = 0; i < n && r == TW_OK && es == hipSuccess; ++i) {
      r = decode_core(c, B, st, 0, nullptr, !fuse_embed);
      if (r == TW_OK) es = launch_sampler(sa, st);
    }
    hipError_t ee = hipStreamEndCapture(st, &g);
    if (r != TW_OK) { if (g) (void)hipGraphDestroy(g); return r; }
    if (es != hipSuccess || ee != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      return fail(c, TW_EHIP, "decode-step graph capture failed: %s", hipGetErrorString(es != hipSuccess ? es : ee));
    }
    hipGraphExec_t ex = nullptr;
    hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) return fail(c, TW_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei));
    c->step_graphs[k] = ex;
    *out = ex;
    return TW_OK;
  };

  // ---- the loop: step s consumes position s and produces the token at position s+1 ----
  static const bool host_timing = getenv("TW_HOST_TIMING") != nullptr;   // where the CALL's wall time goes besides the device loop
  const auto ht0 = std::chrono::steady_clock::now();
  auto ht_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ht0).count(); };
  double ht_first = 0, ht_loop = 0;
  if (n_draft == 0) tic(c, 3, st);
  // two steps per graph launch: measured +2 % at one turbo stream, +0.3 % at 16 large-v3 streams (4 per launch: +3 % / +0.8 %, but
  // up to 7 steps past the last <eos> instead of 5); the finished flags are read LAG launches behind so the host never waits
  static const int group = []() { const char* e = getenv("TW_GRAPH_STEPS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
  // while <eos> is still masked (min_new_tokens not reached) no stream can finish, so nothing is wasted by replaying more steps per
  // launch: 8 at a time there (forced-length decoding, e.g. RTFx runs with a fixed token budget); the reference backend's calls
  // set no minimum and always take the small group
  static const int group_forced = []() { const char* e = getenv("TW_GRAPH_STEPS_FORCED"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
  const int LAG = std::max(1, 4 / group);
  int steps = 0, launches = 0;
  bool all_done = draft_all_done;
  for (int s = s_start; s < max_len - 1 && !all_done;) {
    // step s produces the token at position s + 1, which can be <eos> only once s + 1 - n_prompt >= min_new_tokens
    // (not for the first launch of a call: submitting a graph costs host time in proportion to its nodes, and the GPU is idle until
    // the first one is in; a small first launch covers the submission of the big second one)
    const bool eos_free = use_graph && launches >= 1 && group_forced > group && s + group_forced <= max_len - 1 &&
                          s + group_forced - 1 < n_begin + o->min_new_tokens - 1;
    const int n = eos_free ? group_forced : ((use_graph && s + group <= max_len - 1) ? group : 1);   // steps in this launch
    const int last = s + n - 1;
    const int kb = std::min(((last + 64) / 64) * 64, ((max_len + 63) / 64) * 64);   // keys [0, kb) cover positions s .. last
    if (use_graph) {
      hipGraphExec_t ex = nullptr;
      int r = graph_for(kb, n, &ex);
      if (r != TW_OK) return r;
      HIPCHK(c, hipGraphLaunch(ex, st));
    } else {
      c->dec_key_bound = kb;
      int r = decode_core(c, B, st, 0, nullptr, !fuse_embed);
      if (r != TW_OK) return r;
      HIPCHK(c, launch_sampler(sa, st));
    }
    steps += n;
    s += n;
    const int slot = launches % 8;
    HIPCHK(c, hipMemcpyAsync(c->h_pinned + slot * 64, c->finished, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ring_ev[slot], st));
    if (launches >= LAG) {
      const int ps = (launches - LAG) % 8;
      HIPCHK(c, hipEventSynchronize(c->ring_ev[ps]));
      bool done = true;
      for (int b = 0; b < B; ++b) done &= (c->h_pinned[ps * 64 + b] != 0);
      all_done = done;
    }
    if (launches == 0) ht_first = ht_ms();
    ++launches;
  }
  toc(c, 3, st);
  ht_loop = ht_ms();
  c->last_steps = steps;
  HIPCHK(c, hipMemcpyAsync(hseq, c->seq, sizeof(int) * (size_t)B * P, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (host_timing) {
    float dev_ms = 0.f;
    (void)hipEventElapsedTime(&dev_ms, c->ev0[3], c->ev1[3]);
    fprintf(stderr, "TW_HOST_TIMING generate_greedy B=%d steps=%d launches=%d: first launch submitted at %.3f ms, host loop done at %.3f ms, "
                    "synchronised at %.3f ms; device loop (events) %.3f ms\n", B, steps, launches, ht_first, ht_loop, ht_ms(), dev_ms);
  }

  // common sequence length exactly as HF's loop would have stopped: when the last row hit eos, or at max_len
  const int produced = s_start + steps + 1;  // positions 0 .. s_start + steps are filled
  // first position a SAMPLED token sits at (forced / confirmed draft tokens before it are never <eos>; unconfirmed draft tokens
  // behind `produced` are stale and never looked at)
  const int scan_from = n_draft > 0 ? std::max(s_start, n_begin) : n_prompt;
  int L = scan_from + 1;
  for (int b = 0; b < B; ++b) {
    int lb = produced;
    for (int i = scan_from; i < produced; ++i)
      if (hseq[(size_t)b * P + i] == o->eos_id) { lb = i + 1; break; }
    if (lb > L) L = lb;
  }
  for (int b = 0; b < B; ++b) {
    bool ended = false;
    for (int i = 0; i < out_ld; ++i) {
      int v = o->pad_id;
      if (i < L) {
        v = hseq[(size_t)b * P + i];
        if (ended) v = o->pad_id;
        if (i >= scan_from && v == o->eos_id) ended = true;
      }
      out_ids[(size_t)b * out_ld + i] = v;
    }
  }
  *out_len = L;
  c->last_seq_len = L;
  c->last_n_prompt = n_prompt;
  return TW_OK;
}

// Diagnostics only (NOT part of include/thewhisper.h): copy an internal decoder buffer to the host - tools/dbg/rows_vs_steps.py.
int tw_dbg_copy(tw_ctx* c, const char* what, void* host, size_t bytes) {
  if (!c || !what || !host) return TW_EINVAL;
  TW_ON_DEVICE(c);
  const std::string w(what);
  const void* src = w == "dq" ? c->dq : w == "du" ? (void*)c->du : w == "dstats" ? (void*)c->dstats : w == "datt" ? c->datt : w == "dx0" ? c->dx0 :
                    w == "dx1" ? c->dx1 : w == "dh" ? c->dh : w == "self_k" ? c->self_k : w == "self_v" ? c->self_v : w == "logits" ? (void*)c->logits : nullptr;
  if (!src) return TW_EINVAL;
  (void)hipDeviceSynchronize();
  return hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? TW_OK : TW_EHIP;
}

int tw_last_draft(tw_ctx* c, int32_t* offered, int32_t* accepted, int32_t* launches, int32_t* rounds) {
  if (!c) return TW_EINVAL;
  if (offered) *offered = c->last_draft[0];
  if (accepted) *accepted = c->last_draft[1];
  if (launches) *launches = c->last_draft[2];
  if (rounds) *rounds = c->last_draft[3];
  return TW_OK;
}

int tw_get_alignment(tw_ctx* c, int32_t B, int32_t n_rows, float* out_host, void* stream) {
  if (!c || !out_host) return TW_EINVAL;
  TW_ON_DEVICE(c);
  if (c->Ha == 0) return fail(c, TW_EINVAL, "context has no alignment heads");
  if (B < 1 || B > c->Bmax || n_rows < 1 || n_rows > c->P) return fail(c, TW_EINVAL, "bad B/n_rows");
  hipStream_t st = pick_stream(c, stream);
  const size_t T = c->T, P = c->P, Ha = c->Ha;
  for (int b = 0; b < B; ++b)
    for (size_t h = 0; h < Ha; ++h)
      HIPCHK(c, hipMemcpyAsync(out_host + ((size_t)b * Ha + h) * n_rows * T, c->align + ((size_t)b * Ha + h) * P * T,
                               sizeof(float) * n_rows * T, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return TW_OK;
}

int tw_token_timestamps(tw_ctx* c, int32_t B, int32_t n_prompt, int32_t seq_len, const int32_t* num_frames_host,
                        double time_precision, float* out_ts_host, void* stream) {
  if (!c || !out_ts_host) return fail(c, TW_EINVAL, "tw_token_timestamps: null argument");
  TW_ON_DEVICE(c);
  if (c->Ha == 0) return fail(c, TW_EINVAL, "context has no alignment heads");
  if (B < 1 || B > c->Bmax || seq_len < 2 || seq_len > c->P || n_prompt < 1 || n_prompt >= seq_len)
    return fail(c, TW_EINVAL, "bad B/n_prompt/seq_len (%d,%d,%d)", B, n_prompt, seq_len);
  hipStream_t st = pick_stream(c, stream);
  std::vector<int> cols(B);
  for (int b = 0; b < B; ++b) {
    // HF crops with the Python slice `weights[..., : num_frames // 2]` (HF:models/whisper/generation_whisper.py:346-349):
    // floor division, and a NEGATIVE bound (num_frames - seek < 0 happens in the seek loop for clips shorter than the
    // chunk) counts from the end.  Reproduced literally so token timestamps stay identical to the reference.
    int nc = c->T;
    if (num_frames_host) {
      const int nf = num_frames_host[b];
      nc = (nf >= 0) ? nf / 2 : -((-nf + 1) / 2);   // floor(nf / 2)
      if (nc < 0) nc += c->T;
    }
    if (nc > c->T) nc = c->T;
    if (nc < 0) nc = 0;    // nothing left: HF's DTW on the empty matrix puts every token at time index -1 (dtw_kernel)
    cols[b] = nc;
  }
  HIPCHK(c, hipMemcpy(c->n_cols, cols.data(), sizeof(int) * B, hipMemcpyHostToDevice));
  DtwArgs a{};
  a.align = c->align; a.Ha = c->Ha; a.P = c->P; a.T = c->T; a.B = B; a.n_prompt = n_prompt; a.n_rows = seq_len - 1;
  a.n_cols = c->n_cols; a.median_width = 7; a.zbuf = c->zbuf; a.mat = c->mat; a.trace = c->trace; a.out_ts = c->out_ts;
  // the reference multiplies int64 frame indices by the Python float 0.02 (float64)
  a.time_precision = time_precision;
  tic(c, 4, st);
  HIPCHK(c, launch_token_timestamps(a, st));
  toc(c, 4, st);
  HIPCHK(c, hipMemcpyAsync(out_ts_host, c->out_ts, sizeof(float) * (size_t)B * seq_len, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return TW_OK;
}

int tw_last_timings(tw_ctx* c, float* ms_out5, int32_t* steps_out) {
  if (!c || !ms_out5) return TW_EINVAL;
  TW_ON_DEVICE(c);
  for (int i = 0; i < 5; ++i) {
    ms_out5[i] = -1.f;
    if (c->ev_valid[i]) {
      float ms = 0.f;
      if (hipEventSynchronize(c->ev1[i]) == hipSuccess && hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]) == hipSuccess)
        ms_out5[i] = ms;
    }
  }
  if (steps_out) *steps_out = c->last_steps;
  return TW_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// HIP streams / events owned by THIS library's runtime (the one torch mapped): used by the host-side stage overlap
// (thewhisper_amd/overlap.py) so that no second copy of libamdhip64 is ever dlopen'ed by name.
// ---------------------------------------------------------------------------------------------
extern "C" {

int tw_vad_energy(int32_t device, const float* pcm_dev, int64_t stream_stride, int32_t B, int32_t n_frames, float* state_dev,
                  float* prob_dev, void* stream) {
  if (!pcm_dev || !state_dev || !prob_dev || B < 1 || n_frames < 1 || stream_stride < (int64_t)n_frames * 512) return TW_EINVAL;
  if ((reinterpret_cast<uintptr_t>(pcm_dev) & 15) || (stream_stride & 3)) return TW_EINVAL;   // 16-byte frame loads
  DeviceGuard guard(device);
  hipError_t e = launch_vad_energy(pcm_dev, stream_stride, B, n_frames, state_dev, prob_dev, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { g_create_error = std::string("vad_energy_kernel: ") + hipGetErrorString(e); return TW_EHIP; }
  return TW_OK;
}

int tw_stream_create_masked(int32_t device, const uint32_t* cu_mask, int32_t n_words, void** out_stream) {
  if (!cu_mask || n_words < 1 || !out_stream) return TW_EINVAL;
  DeviceGuard guard(device);
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask);
  if (e != hipSuccess) { g_create_error = std::string("hipExtStreamCreateWithCUMask: ") + hipGetErrorString(e); return TW_EHIP; }
  *out_stream = st;
  return TW_OK;
}

int tw_stream_destroy(void* stream) {
  if (!stream) return TW_OK;
  (void)hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));
  return hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? TW_OK : TW_EHIP;
}

int tw_stream_synchronize(void* stream) {
  return hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? TW_OK : TW_EHIP;
}

int tw_event_create(int32_t device, void** out_event) {
  if (!out_event) return TW_EINVAL;
  DeviceGuard guard(device);
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return TW_EHIP;
  *out_event = ev;
  return TW_OK;
}

int tw_event_destroy(void* event) {
  if (!event) return TW_OK;
  return hipEventDestroy(reinterpret_cast<hipEvent_t>(event)) == hipSuccess ? TW_OK : TW_EHIP;
}

int tw_event_record(void* event, void* stream) {
  return hipEventRecord(reinterpret_cast<hipEvent_t>(event), reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? TW_OK : TW_EHIP;
}

int tw_stream_wait_event(void* stream, void* event) {
  return hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(event), 0) == hipSuccess ? TW_OK : TW_EHIP;
}

}  // extern "C"
