// C-ABI of libstlt_hip.so (see include/stlt_hip.h): argument checks, error string, per-kernel event
// timing, and the whole-path orchestration (StltBackbone.forward / Stlt.forward as a fixed launch sequence
// on the caller's stream — no allocation, no synchronisation, graph-capturable).
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

namespace {
thread_local char g_err[512] = "";

// Per-kernel timing: the switch is process-wide, the records and the event pool are per device (an event belongs to the
// device it was created on), and a launch's two events are recorded on the launch's own stream.
struct ProfRec { int kid; hipEvent_t a, b; double flops, bytes; int kernels; char note[STLT_PROF_NOTE]; };
thread_local char t_note[STLT_PROF_NOTE] = "";
thread_local double t_flops = 0.0, t_bytes = 0.0;
thread_local int t_kernels = 0;  // kernel launches checked (stlt_check_launch) inside the open scope
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs[STLT_MAX_DEVICES];
std::vector<hipEvent_t> g_event_pool[STLT_MAX_DEVICES];
thread_local hipEvent_t g_open_start = nullptr;
thread_local int g_scope_depth = 0;
double g_gemm_flops[STLT_MAX_DEVICES] = {};

hipEvent_t get_event(int dev) {
  auto& pool = g_event_pool[dev];
  if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
StltPerDeviceInt g_cus;
}  // namespace

int stlt_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev;
}

int stlt_device_cus() {
  int& n = g_cus.ref();
  if (n == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, stlt_current_device()) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

int stlt_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int stlt_check_launch(const char* what) {
  if (g_scope_depth > 0) ++t_kernels;
  else if (g_prof_on) {  // a kernel outside every scope still gets a record (no events: 0 us), so that the records cover every kernel the library starts
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec rec{STLT_K_MISC, nullptr, nullptr, 0.0, 0.0, 1, {0}};
    snprintf(rec.note, sizeof(rec.note), "%s (outside the recorder's scopes)", what);
    g_prof_recs[stlt_current_device() & (STLT_MAX_DEVICES - 1)].push_back(rec);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return stlt_set_error((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

void stlt_prof_begin(int kid, hipStream_t s) {
  if (g_scope_depth++ > 0) return;  // nested launcher: the outer scope times it
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_open_start = get_event(stlt_current_device() & (STLT_MAX_DEVICES - 1));
  t_note[0] = 0;
  t_flops = t_bytes = 0.0;
  t_kernels = 0;
  (void)hipEventRecord(g_open_start, s);
}

void stlt_prof_add_flops(double flops) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_gemm_flops[stlt_current_device() & (STLT_MAX_DEVICES - 1)] += flops;
  t_flops += flops;
}

// What the open (outermost) scope's launch is: shape, tile, workgroups, rounds, k-steps — appended to the launch's record for
// stlt_prof_launches (tools/launch_bound.py prices every launch of a step against its own bound).  Free when the recorder is off.
void stlt_prof_note(const char* fmt, ...) {
  if (!g_prof_on || g_scope_depth == 0 || !g_open_start) return;
  const size_t used = strlen(t_note);
  if (used + 2 >= sizeof(t_note)) return;
  if (used) { t_note[used] = ' '; t_note[used + 1] = 0; }
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_note + strlen(t_note), sizeof(t_note) - strlen(t_note), fmt, ap);
  va_end(ap);
}
void stlt_prof_add_bytes(double bytes) {
  if (g_prof_on && g_scope_depth > 0) t_bytes += bytes;
}
// FLOPs of a launch that is not a matrix-core product of the GEMM roofline (fused MHSA, attention cores): on the launch's record only
void stlt_prof_note_flops(double flops) {
  if (g_prof_on && g_scope_depth > 0) t_flops += flops;
}

void stlt_prof_end(int kid, hipStream_t s) {
  if (--g_scope_depth > 0) return;
  if (!g_prof_on || !g_open_start) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const int dev = stlt_current_device() & (STLT_MAX_DEVICES - 1);
  hipEvent_t b = get_event(dev);
  (void)hipEventRecord(b, s);
  ProfRec rec{kid, g_open_start, b, t_flops, t_bytes, t_kernels, {0}};
  memcpy(rec.note, t_note, sizeof(rec.note));
  g_prof_recs[dev].push_back(rec);
  g_open_start = nullptr;
}

unsigned long long* g_stlt_debug_buf = nullptr;

// Layers whose sequences have at most 64 tokens (and 64-channel heads) run the in-projection and the attention core as ONE kernel
// (mhsa.hip: the packed QKV tensor never reaches HBM).  Round 3 had this for 32-frame clips only; round 4's kernel takes the
// reference's real layouts (T = 17 / 33, datasets.py:97-113), cfg4's 64 frames and the training forward.  STLT_FUSED_MHSA=0
// restores the two launches everywhere, STLT_FUSED_MHSA_SPATIAL=0 for the spatial tower's (non-causal) layers only (A/B runs).
bool stlt_fused_mhsa_on(int causal) {
  static const bool on = [] { const char* e = getenv("STLT_FUSED_MHSA"); return e ? atoi(e) != 0 : true; }();
  static const bool spatial = [] { const char* e = getenv("STLT_FUSED_MHSA_SPATIAL"); return e ? atoi(e) != 0 : true; }();
  return on && (causal || spatial);
}

extern "C" {

int stlt_debug_set_buffer(void* dev_buf) { g_stlt_debug_buf = (unsigned long long*)dev_buf; return 0; }
size_t stlt_debug_buffer_bytes(void) { return ((size_t)54 * (size_t)stlt_device_cus() + 1024) * sizeof(unsigned long long); }

int stlt_version(void) { return STLT_VERSION; }
const char* stlt_last_error(void) { return g_err; }

int stlt_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return 0;
}

double stlt_prof_take_gemm_flops(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double& f = g_gemm_flops[stlt_current_device() & (STLT_MAX_DEVICES - 1)];
  const double out = f;
  f = 0.0;
  return out;
}

int stlt_prof_collect(double* ms_out, int64_t* launches_out) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int k = 0; k < STLT_K_COUNT; ++k) { if (ms_out) ms_out[k] = 0.0; if (launches_out) launches_out[k] = 0; }
  const int dev = stlt_current_device() & (STLT_MAX_DEVICES - 1);  // the records of the device that is current
  for (auto& r : g_prof_recs[dev]) {
    if (!r.a) continue;  // a kernel outside the scopes: counted by stlt_prof_launches only
    hipError_t e = hipEventSynchronize(r.b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return stlt_set_error((int)e, "stlt_prof_collect: %s", hipGetErrorString(e));
    if (r.kid >= 0 && r.kid < STLT_K_COUNT) { if (ms_out) ms_out[r.kid] += ms; if (launches_out) launches_out[r.kid] += 1; }
    g_event_pool[dev].push_back(r.a);
    g_event_pool[dev].push_back(r.b);
  }
  g_prof_recs[dev].clear();
  return 0;
}

int stlt_prof_launches(stlt_prof_launch* out, int64_t cap, int64_t* n_out) {
  if (!n_out || (cap > 0 && !out)) return stlt_set_error(STLT_EINVAL, "stlt_prof_launches: null argument");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const int dev = stlt_current_device() & (STLT_MAX_DEVICES - 1);
  int64_t n = 0;
  for (auto& r : g_prof_recs[dev]) {
    hipError_t e = hipSuccess;
    float ms = 0.f;
    if (r.a) {
      e = hipEventSynchronize(r.b);
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
    }
    if (e != hipSuccess) return stlt_set_error((int)e, "stlt_prof_launches: %s", hipGetErrorString(e));
    if (n < cap) {
      out[n].kid = r.kid; out[n].kernels = r.kernels; out[n].us = ms * 1e3f; out[n].flops = r.flops; out[n].bytes = r.bytes;
      memcpy(out[n].note, r.note, sizeof(out[n].note));
      out[n].note[sizeof(out[n].note) - 1] = 0;
    }
    ++n;
    if (r.a) { g_event_pool[dev].push_back(r.a); g_event_pool[dev].push_back(r.b); }
  }
  g_prof_recs[dev].clear();
  *n_out = n;
  return 0;
}

int stlt_embed_fwd(const int64_t* categories, const float* boxes, const float* scores, const float* cat_table,
                   int64_t n_categories, const float* box_w, const float* box_b, const float* score_w,
                   const float* score_b, const float* ln_w, const float* ln_b, float eps, int64_t n_tokens, int64_t d,
                   float* out, stlt_stream_t stream) {
  return launch_embed(categories, boxes, scores, cat_table, n_categories, box_w, box_b, score_w, score_b, ln_w, ln_b,
                      eps, n_tokens, d, out, (hipStream_t)stream);
}

int stlt_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int64_t M,
                    int64_t N, int64_t K, int act, stlt_stream_t stream) {
  return launch_linear(x, ldx, w, bias, y, ldy, M, N, K, act, (hipStream_t)stream);
}

int stlt_linear_small_fwd(const float* x, int64_t ldx, const float* w, const float* bias, const float* r, int64_t ldr, float* y, int64_t ldy, int64_t M,
                          int64_t N, int64_t K, int act, int tile, stlt_stream_t stream) {
  const int code = stlt_gemm16_tile_from_public(tile);
  if (code == 0) return stlt_set_error(STLT_EINVAL, "stlt_linear_small_fwd: tile = columns | rows << 16: 128 rows (or 0) x {48,64,96,128,144,192}, 64 x {64,96,128,160,192,256}, 32 x {128,192,256}");
  bool taken = false;
  if (int e = launch_linear_gemm16(x, ldx, w, K, bias, r, ldr, y, ldy, M, N, K, act, (hipStream_t)stream, &taken, code)) return e;
  return taken || M == 0 ? 0 : stlt_set_error(STLT_EINVAL, "stlt_linear_small_fwd: shape or activation not taken by the small-tile kernel");
}

int stlt_input_grad_small(const float* dy, int64_t ld_dy, const float* w, int64_t n_out, int64_t k_in, const float* r, int64_t ldr, float* dx, int64_t ld_dx,
                          int64_t M, int tile, stlt_ctx* ctx, stlt_stream_t stream) {
  // tile 0: routed by the launch-time estimate, and — when `ctx` holds a current transposed copy of w — run as a forward product on the copy
  const int code = tile == 0 ? 0 : stlt_gemm16_tile_from_public(tile);
  if (tile != 0 && code == 0) return stlt_set_error(STLT_EINVAL, "stlt_input_grad_small: tile = 0 (by estimate) or columns | rows << 16: 128 rows (or 0) x {48,64,96,128,144,192}, 64 x {64,96,128,160,192,256}, 32 x {128,192,256}");
  StltCtxScope ctx_scope(ctx, (hipStream_t)stream);
  if (ctx_scope.error()) return ctx_scope.error();
  bool taken = false;
  if (int e = launch_input_grad_gemm16(dy, ld_dy, w, n_out, k_in, r, ldr, dx, ld_dx, M, (hipStream_t)stream, &taken, code)) return e;
  return taken || M == 0 ? 0 : stlt_set_error(STLT_EINVAL, "stlt_input_grad_small: shape not taken by the small-tile kernel");
}

int stlt_linear_small_choice(int64_t M, int64_t N, int64_t K) { return stlt_gemm16_tile_to_public(stlt_gemm16_choice(M, N, K, K, K)); }
int stlt_input_grad_small_choice(int64_t M, int64_t n_out, int64_t k_in) { return stlt_gemm16_tile_to_public(stlt_gemm16_choice(M, k_in, n_out, n_out, k_in, true)); }
int stlt_set_gemm_small_tiles(int mode) { return stlt_gemm16_set_mode(mode); }

int stlt_gemm(int transA, int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* r, int64_t ldr,
              float* c, int64_t ldc, int64_t slab_stride, int64_t M, int64_t N, int64_t K, int n_split,
              stlt_stream_t stream) {
  return launch_gemm(transA, transB, a, lda, b, ldb, nullptr, r, ldr, c, ldc, slab_stride, M, N, K, n_split,
                     STLT_ACT_NONE, (hipStream_t)stream);
}

int stlt_weight_grad_group(const stlt_wgrad_item* items, int n_items, stlt_stream_t stream) {
  static_assert(sizeof(stlt_wgrad_item) == sizeof(StltWeightGradItem), "the C-ABI item is the internal item");
  return launch_weight_grad_group(reinterpret_cast<const StltWeightGradItem*>(items), n_items, (hipStream_t)stream);
}

size_t stlt_gemm_scratch_bytes(void) { return STLT_GEMM_SCRATCH_BYTES; }

int stlt_gemm_set_scratch(void* scratch, size_t bytes) {
  if (scratch && bytes < STLT_GEMM_SCRATCH_BYTES)
    return stlt_set_error(STLT_EWORKSPACE, "gemm scratch %zu B < required %zu B", bytes, (size_t)STLT_GEMM_SCRATCH_BYTES);
  stlt_gemm_set_scratch_impl(scratch, bytes);
  return 0;
}

int stlt_reduce_slabs(const float* slabs, int64_t stride, int n_slabs, float* dst, int64_t n, int accumulate,
                      stlt_stream_t stream) {
  return launch_reduce_slabs(slabs, stride, n_slabs, dst, n, accumulate, (hipStream_t)stream);
}

int stlt_attn_core_fwd(const float* qkv, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, int64_t dh,
                       float* ctx, stlt_stream_t stream) {
  return launch_attn(qkv, kpm, causal, S, L, H, dh, ctx, causal ? STLT_K_ATTN_TEMPORAL : STLT_K_ATTN_SPATIAL,
                     (hipStream_t)stream);
}

int stlt_fused_mhsa_active(int64_t T, int64_t d, int64_t H) { return (stlt_fused_mhsa_on(1) && stlt_mhsa_fused_pays(1024, T, H, d, 1)) ? 1 : 0; }
int stlt_fused_mhsa_used(int64_t S, int64_t L, int64_t d, int64_t H, int causal) { return (stlt_fused_mhsa_on(causal) && stlt_mhsa_fused_pays(S, L, H, d, causal)) ? 1 : 0; }

int stlt_mhsa_fused_fwd(const float* x, const float* in_proj_w, const float* in_proj_b, const uint8_t* kpm, int64_t S, int64_t L, int64_t H,
                        int64_t d, float* ctx, stlt_stream_t stream) {
  return launch_mhsa_fused(x, in_proj_w, in_proj_b, kpm, S, L, H, d, ctx, (hipStream_t)stream);
}

int stlt_mhsa_fused_fwd_ex(const float* x, const float* in_proj_w, const float* in_proj_b, const uint8_t* kpm, int causal, int64_t S, int64_t L,
                           int64_t H, int64_t d, float dropout_p, uint64_t seed, uint32_t site, float* ctx, float* qkv_out, stlt_stream_t stream) {
  if (dropout_p < 0.f || dropout_p >= 1.f) return stlt_set_error(STLT_EINVAL, "stlt_mhsa_fused_fwd_ex: dropout_p must be in [0, 1)");
  return launch_mhsa_fused(x, in_proj_w, in_proj_b, kpm, S, L, H, d, ctx, (hipStream_t)stream, causal, qkv_out, stlt_drop_make(dropout_p, seed), site);
}

int stlt_attn_cross_fwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kpm, int causal,
                        int64_t S, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float* ctx, stlt_stream_t stream) {
  return launch_attn_general(q, ldq, k, v, ldkv, kpm, causal, S, Lq, Lk, H, dh, ctx,
                             causal ? STLT_K_ATTN_TEMPORAL : STLT_K_ATTN_SPATIAL, (hipStream_t)stream);
}

int stlt_attn_ragged_fwd(const float* qkv, const int32_t* seg_start, const int32_t* seg_end, int causal, int64_t M, int64_t H,
                         int64_t dh, float* ctx, stlt_stream_t stream) {
  return launch_attn_ragged(qkv, seg_start, seg_end, causal, M, H, dh, ctx, causal ? STLT_K_ATTN_TEMPORAL : STLT_K_ATTN_SPATIAL,
                            (hipStream_t)stream);
}

int stlt_add_layernorm_fwd(const float* x, int64_t ldx, const float* res, int64_t ldres, const float* ln_w,
                           const float* ln_b, float eps, int64_t M, int64_t d, float* out, int64_t ldout,
                           stlt_stream_t stream) {
  return launch_add_layernorm(x, ldx, res, ldres, ln_w, ln_b, eps, M, d, out, ldout, (hipStream_t)stream);
}

int stlt_frames_embed_fwd(const float* spatial, int64_t row_stride, const int64_t* frame_types, const float* pos_table,
                          const float* type_table, const float* ln_w, const float* ln_b, float eps, int64_t B,
                          int64_t T, int64_t d, float* out, stlt_stream_t stream) {
  return launch_frames_embed(spatial, row_stride, frame_types, pos_table, type_table, ln_w, ln_b, eps, B, T, d, out,
                             (hipStream_t)stream);
}

int stlt_gather_last_fwd(const float* x, const int64_t* lengths, int64_t B, int64_t T, int64_t d, float* out,
                         stlt_stream_t stream) {
  return launch_gather_last(x, lengths, B, T, d, out, (hipStream_t)stream);
}

// ------------------------------------------------------------------ whole path

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
  size_t x, x1, qkv, ctx, tmp, hh, head, sk, ridx, total;
};

static WsLayout ws_layout(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_classes) {
  const size_t tok = (size_t)B * T * N;  // spatial tokens >= temporal tokens
  const size_t f = sizeof(float);
  WsLayout w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  w.x = take(tok * d * f);
  w.x1 = take(tok * d * f);
  w.qkv = take(tok * 3 * d * f);
  w.ctx = take(tok * d * f);
  w.tmp = take(tok * d * f);
  w.hh = take(tok * 4 * d * f);
  w.head = take((size_t)B * (3 * d + n_classes) * f);
  w.sk = take(STLT_GEMM_SCRATCH_BYTES);  // stream-K partial tiles of under-filled GEMM launches
  w.ridx = take(ragged_index_bytes(B, T, N));  // STLT_FLAG_SKIP_PADDING: index of the real tokens / frames
  w.total = off;
  return w;
}

size_t stlt_workspace_bytes(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_classes) {
  if (B <= 0 || T <= 0 || N <= 0 || d <= 0) return 0;
  return ws_layout(B, T, N, d, n_classes < 0 ? 0 : n_classes).total;
}

#define TRY(expr) do { int _e = (expr); if (_e) return _e; } while (0)

// The residual adds of a post-norm layer ride in the out-proj / FFN2 epilogues and the LayerNorm passes read one tensor
// instead of two (bit-identical logits: the accumulators start from the bias, the residual is added last).  Round 2 measured
// it a wash (the residual tile was read inside the exposed epilogue: LayerNorm 3.06 -> 2.00 ms, GEMMs +1.0 ms); with the
// first half of a tile's residual pieces requested during the tile's last k-step (gemm.hip: STLT_GEMM_RES_PREFETCH) the GEMMs
// pay 0.4 ms and the forward gains 0.5 ms at cfg2 / 1024 clips (profiles/round3_fwd_residual_ab.txt).  STLT_FUSE_RESIDUAL=0
// restores the separate pass (A/B runs).
static bool fuse_residual() {
  static const bool on = [] { const char* e = getenv("STLT_FUSE_RESIDUAL"); return e ? atoi(e) != 0 : true; }();
  return on;
}

// qkv projection + attention core of a layer: ctx (M,d) from x (M,d)
static int qkv_attention(const stlt_layer_params& lp, int64_t d, int64_t H, const float* x, int64_t M, int64_t S, int64_t L,
                         const uint8_t* kpm, int causal, int kid, float* qkv, float* ctx, hipStream_t s) {
  // with the opt-in split-bf16 products on, a temporal in-projection they take is faster as its own launch (+ the attention core)
  // than inside the fused f32-MFMA kernel
  if (stlt_fused_mhsa_on(causal) && stlt_mhsa_fused_pays(S, L, H, d, causal) && !stlt_split_bf16_takes(M, 3 * d, d, d, d))
    return launch_mhsa_fused(x, lp.in_proj_w, lp.in_proj_b, kpm, S, L, H, d, ctx, s, causal);
  if (int e = launch_linear(x, d, lp.in_proj_w, lp.in_proj_b, qkv, 3 * d, M, 3 * d, d, STLT_ACT_NONE, s)) return e;
  return launch_attn(qkv, kpm, causal, S, L, H, d / H, ctx, kid, s);
}

// One post-norm encoder layer (nn.TransformerEncoderLayer as configured at models.py:46-52,118-124) on M
// compact rows of width d.  `out` may alias `x` (x is last read by the norm1 residual); x1 must not.
static int encoder_layer(const stlt_layer_params& lp, int64_t d, int64_t H, const float* x, int64_t M, int64_t S,
                         int64_t L, const uint8_t* kpm, int causal, int kid, float* qkv, float* ctx, float* tmp,
                         float* x1, float* hh, float* out, hipStream_t s) {
  TRY(qkv_attention(lp, d, H, x, M, S, L, kpm, causal, kid, qkv, ctx, s));
  if (fuse_residual()) {  // the residual adds ride in the out-proj / FFN2 epilogues: the norm passes read one tensor
    TRY(launch_linear_add(ctx, d, lp.out_proj_w, lp.out_proj_b, x, d, tmp, d, M, d, d, s));
    TRY(launch_add_layernorm(tmp, d, nullptr, 0, lp.norm1_w, lp.norm1_b, 1e-5f, M, d, x1, d, s));
    TRY(launch_linear(x1, d, lp.lin1_w, lp.lin1_b, hh, 4 * d, M, 4 * d, d, STLT_ACT_GELU, s));
    TRY(launch_linear_add(hh, 4 * d, lp.lin2_w, lp.lin2_b, x1, d, tmp, d, M, d, 4 * d, s));
    TRY(launch_add_layernorm(tmp, d, nullptr, 0, lp.norm2_w, lp.norm2_b, 1e-5f, M, d, out, d, s));
    return 0;
  }
  TRY(launch_linear(ctx, d, lp.out_proj_w, lp.out_proj_b, tmp, d, M, d, d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(tmp, d, x, d, lp.norm1_w, lp.norm1_b, 1e-5f, M, d, x1, d, s));
  TRY(launch_linear(x1, d, lp.lin1_w, lp.lin1_b, hh, 4 * d, M, 4 * d, d, STLT_ACT_GELU, s));
  TRY(launch_linear(hh, 4 * d, lp.lin2_w, lp.lin2_b, tmp, d, M, d, 4 * d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(tmp, d, x1, d, lp.norm2_w, lp.norm2_b, 1e-5f, M, d, out, d, s));
  return 0;
}

static int check_params(const stlt_params* p, const stlt_inputs* in, bool need_head) {
  if (!p || !in) return stlt_set_error(STLT_EINVAL, "null params/inputs");
  if (!stlt_heads_ok(p->d, p->H))
    return stlt_set_error(STLT_EINVAL, "hidden_size %lld / heads %lld: need hidden_size %% heads == 0, a head dim of at most 256 and hidden_size %% 4 == 0", (long long)p->d, (long long)p->H);
  if (in->B <= 0 || in->T <= 0 || in->N <= 0) return stlt_set_error(STLT_EINVAL, "empty batch (B=%lld,T=%lld,N=%lld)", (long long)in->B, (long long)in->T, (long long)in->N);
  if (in->T > p->n_positions) return stlt_set_error(STLT_EINVAL, "T=%lld exceeds the position table (%lld rows)", (long long)in->T, (long long)p->n_positions);
  if (!in->categories || !in->boxes || !in->kpm_boxes || !in->frame_types || !in->kpm_frames)
    return stlt_set_error(STLT_EINVAL, "null input tensor");
  if (p->n_spatial < 0 || p->n_temporal < 0 || (p->n_spatial && !p->spatial) || (p->n_temporal && !p->temporal))
    return stlt_set_error(STLT_EINVAL, "layer tables missing");
  if (need_head && (!p->fc1_w || !p->fc1_b || !p->head_ln_w || !p->head_ln_b || !p->fc2_w || !p->fc2_b || !in->lengths || p->n_classes <= 0))
    return stlt_set_error(STLT_EINVAL, "prediction head parameters / lengths missing");
  return 0;
}

}  // extern "C"

// Backbone body.  With last_rows != nullptr the final temporal layer only produces the rows the head reads
// (Stlt.forward, models.py:189-192: out[lengths-1, arange(B)]) into last_rows (B,d): K/V/Q are projected for every
// frame, the attention core runs as usual, but out-proj / norm1 / FFN / norm2 run on the B gathered rows only.
static int backbone_impl(const stlt_params* p, const stlt_inputs* in, void* workspace, size_t workspace_bytes,
                         int flags, float* out_btd, float* last_rows, hipStream_t s) {
  const int64_t B = in->B, T = in->T, N = in->N, d = p->d, H = p->H;
  const WsLayout w = ws_layout(B, T, N, d, p->n_classes < 0 ? 0 : p->n_classes);
  if (!workspace || workspace_bytes < w.total)
    return stlt_set_error(STLT_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.total);
  char* base = (char*)workspace;
  StltGemmScratch gemm_scratch(base + w.sk, STLT_GEMM_SCRATCH_BYTES);
  float* x = (float*)(base + w.x);
  float* x1 = (float*)(base + w.x1);
  float* qkv = (float*)(base + w.qkv);
  float* ctx = (float*)(base + w.ctx);
  float* tmp = (float*)(base + w.tmp);
  float* hh = (float*)(base + w.hh);

  const int64_t tok = B * T * N, BT = B * T;
  // K1
  TRY(launch_embed(in->categories, in->boxes, in->scores, p->cat_emb, p->n_categories, p->box_w, p->box_b, p->score_w,
                   p->score_b, p->emb_ln_w, p->emb_ln_b, p->ln_eps, tok, d, x, s));
  // spatial transformer: sequences = frames (B*T), tokens = objects (N), key-padding mask only
  const bool cls_only = (flags & STLT_FLAG_CLS_ONLY_LAST_SPATIAL) && p->n_spatial > 0 && N > 1;
  const int64_t full_layers = cls_only ? p->n_spatial - 1 : p->n_spatial;
  for (int64_t l = 0; l < full_layers; ++l)
    TRY(encoder_layer(p->spatial[l], d, H, x, tok, BT, N, in->kpm_boxes, 0, STLT_K_ATTN_SPATIAL, qkv, ctx, tmp, x1, hh, x, s));
  const float* cls_rows = x;
  int64_t cls_stride = N * d;
  if (cls_only) {
    // Only token 0 of each frame is read after the spatial transformer (models.py:79).  K and V are needed
    // for every object token, Q / out-proj / FFN only for the B*T CLS rows (rows of stride N*d in x).
    const stlt_layer_params& lp = p->spatial[p->n_spatial - 1];
    TRY(launch_linear(x, d, lp.in_proj_w + d * d, lp.in_proj_b + d, qkv + d, 3 * d, tok, 2 * d, d, STLT_ACT_NONE, s));
    TRY(launch_linear(x, N * d, lp.in_proj_w, lp.in_proj_b, qkv, N * 3 * d, BT, d, d, STLT_ACT_NONE, s));
    // non-CLS query rows of qkv hold stale scratch; each attention output row depends on its own query row
    // only, and only the CLS rows of ctx are read below.
    TRY(launch_attn(qkv, in->kpm_boxes, 0, BT, N, H, d / H, ctx, STLT_K_ATTN_SPATIAL, s));
    TRY(launch_linear(ctx, N * d, lp.out_proj_w, lp.out_proj_b, tmp, d, BT, d, d, STLT_ACT_NONE, s));
    TRY(launch_add_layernorm(tmp, d, x, N * d, lp.norm1_w, lp.norm1_b, 1e-5f, BT, d, x1, d, s));
    TRY(launch_linear(x1, d, lp.lin1_w, lp.lin1_b, hh, 4 * d, BT, 4 * d, d, STLT_ACT_GELU, s));
    TRY(launch_linear(hh, 4 * d, lp.lin2_w, lp.lin2_b, tmp, d, BT, d, 4 * d, STLT_ACT_NONE, s));
    TRY(launch_add_layernorm(tmp, d, x1, d, lp.norm2_w, lp.norm2_b, 1e-5f, BT, d, ctx, d, s));
    cls_rows = ctx;
    cls_stride = d;
  }
  // K7: CLS select + position + frame type + LN -> (B,T,d).  x1 is free here (its last reader was the
  // final norm2 above), x / ctx still hold the CLS rows being read.
  float* tbuf = p->n_temporal > 0 ? x1 : out_btd;
  TRY(launch_frames_embed(cls_rows, cls_stride, in->frame_types, p->pos_emb, p->type_emb, p->frames_ln_w,
                          p->frames_ln_b, p->ln_eps, B, T, d, tbuf, s));
  // temporal transformer: sequences = clips (B), tokens = frames (T), causal + key padding.  Layers run in
  // place on tbuf (a layer's input is last read by its norm1), x is the post-norm1 scratch; the last layer
  // writes the caller's buffer.
  const int64_t n_full = last_rows ? p->n_temporal - 1 : p->n_temporal;
  for (int64_t l = 0; l < n_full; ++l) {
    float* dst = (l == p->n_temporal - 1) ? out_btd : tbuf;
    TRY(encoder_layer(p->temporal[l], d, H, tbuf, BT, B, T, in->kpm_frames, 1, STLT_K_ATTN_TEMPORAL, qkv, ctx, tmp,
                      x, hh, dst, s));
  }
  if (last_rows) {
    const stlt_layer_params& lp = p->temporal[p->n_temporal - 1];
    float* g_ctx = x;                      // (B,d) gathered attention rows
    float* g_res = x + (size_t)B * d;      // (B,d) gathered layer-input rows (residual)
    float* g_x1 = hh + (size_t)B * 4 * d;  // (B,d) post-norm1, behind the (B,4d) FFN hidden (T > 1 => hh holds >= 2*B*4d)
    TRY(qkv_attention(lp, d, H, tbuf, BT, B, T, in->kpm_frames, 1, STLT_K_ATTN_TEMPORAL, qkv, ctx, s));
    TRY(launch_gather_last(ctx, in->lengths, B, T, d, g_ctx, s));
    TRY(launch_gather_last(tbuf, in->lengths, B, T, d, g_res, s));
    TRY(launch_linear(g_ctx, d, lp.out_proj_w, lp.out_proj_b, tmp, d, B, d, d, STLT_ACT_NONE, s));
    TRY(launch_add_layernorm(tmp, d, g_res, d, lp.norm1_w, lp.norm1_b, 1e-5f, B, d, g_x1, d, s));
    TRY(launch_linear(g_x1, d, lp.lin1_w, lp.lin1_b, hh, 4 * d, B, 4 * d, d, STLT_ACT_GELU, s));
    TRY(launch_linear(hh, 4 * d, lp.lin2_w, lp.lin2_b, tmp, d, B, d, 4 * d, STLT_ACT_NONE, s));
    TRY(launch_add_layernorm(tmp, d, g_x1, d, lp.norm2_w, lp.norm2_b, 1e-5f, B, d, last_rows, d, s));
  }
  return 0;
}

// One encoder layer over M compacted rows cut into variable-length segments (ragged.hip): same arithmetic as
// encoder_layer, attention restricted to the row's own segment (no padded keys exist).
static int encoder_layer_ragged(const stlt_layer_params& lp, int64_t d, int64_t H, const float* x, int64_t M, const int* seg_start,
                                const int* seg_end, int causal, int kid, float* qkv, float* ctx, float* tmp, float* x1,
                                float* hh, float* out, hipStream_t s) {
  TRY(launch_linear(x, d, lp.in_proj_w, lp.in_proj_b, qkv, 3 * d, M, 3 * d, d, STLT_ACT_NONE, s));
  TRY(launch_attn_ragged(qkv, seg_start, seg_end, causal, M, H, d / H, ctx, kid, s));
  if (fuse_residual()) {  // the residual adds ride in the out-proj / FFN2 epilogues: the norm passes read one tensor
    TRY(launch_linear_add(ctx, d, lp.out_proj_w, lp.out_proj_b, x, d, tmp, d, M, d, d, s));
    TRY(launch_add_layernorm(tmp, d, nullptr, 0, lp.norm1_w, lp.norm1_b, 1e-5f, M, d, x1, d, s));
    TRY(launch_linear(x1, d, lp.lin1_w, lp.lin1_b, hh, 4 * d, M, 4 * d, d, STLT_ACT_GELU, s));
    TRY(launch_linear_add(hh, 4 * d, lp.lin2_w, lp.lin2_b, x1, d, tmp, d, M, d, 4 * d, s));
    TRY(launch_add_layernorm(tmp, d, nullptr, 0, lp.norm2_w, lp.norm2_b, 1e-5f, M, d, out, d, s));
    return 0;
  }
  TRY(launch_linear(ctx, d, lp.out_proj_w, lp.out_proj_b, tmp, d, M, d, d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(tmp, d, x, d, lp.norm1_w, lp.norm1_b, 1e-5f, M, d, x1, d, s));
  TRY(launch_linear(x1, d, lp.lin1_w, lp.lin1_b, hh, 4 * d, M, 4 * d, d, STLT_ACT_GELU, s));
  TRY(launch_linear(hh, 4 * d, lp.lin2_w, lp.lin2_b, tmp, d, M, d, 4 * d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(tmp, d, x1, d, lp.norm2_w, lp.norm2_b, 1e-5f, M, d, out, d, s));
  return 0;
}

// The tail of an encoder layer (out-proj, norm1, FFN, norm2) on n rows picked out of the layer's attention output
// and input: what the layer computes for rows whose output is the only thing read afterwards.
static int encoder_tail_rows(const stlt_layer_params& lp, int64_t d, const float* ctx, const float* x, const int* rows, int64_t n,
                             float* g_ctx, float* g_res, float* g_x1, float* tmp, float* hh, float* out, hipStream_t s) {
  TRY(launch_gather_rows(ctx, d, rows, n, d, g_ctx, s));
  TRY(launch_gather_rows(x, d, rows, n, d, g_res, s));
  TRY(launch_linear(g_ctx, d, lp.out_proj_w, lp.out_proj_b, tmp, d, n, d, d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(tmp, d, g_res, d, lp.norm1_w, lp.norm1_b, 1e-5f, n, d, g_x1, d, s));
  TRY(launch_linear(g_x1, d, lp.lin1_w, lp.lin1_b, hh, 4 * d, n, 4 * d, d, STLT_ACT_GELU, s));
  TRY(launch_linear(hh, 4 * d, lp.lin2_w, lp.lin2_b, tmp, d, n, d, 4 * d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(tmp, d, g_x1, d, lp.norm2_w, lp.norm2_b, 1e-5f, n, d, out, d, s));
  return 0;
}

// Stlt.forward up to the rows the head reads, computed on the real tokens / frames only (STLT_FLAG_SKIP_PADDING).
// Synchronises the stream once, to read the two row counts the launches are sized by.
// out_btd != null (the fusion models' layout branch): every temporal layer runs on all real frames and the result is
// scattered into the (B,T,d) tensor, padded frames zero — those rows are only ever masked keys or queries whose outputs
// nobody reads (models.py:403-431, 470), so the logits are the padded schedule's; h0 is not produced.
static int forward_ragged(const stlt_params* p, const stlt_inputs* in, void* workspace, const WsLayout& w, float* h0, float* out_btd,
                          hipStream_t s) {
  const int64_t B = in->B, T = in->T, N = in->N, d = p->d, H = p->H;
  // head dims other than 64 run the ragged attention on attn_any.hip, whose forward holds at most 1024 keys of a segment (a frame's N
  // slots, a clip's T frames) in LDS: longer segments are refused here instead of being truncated in the kernel
  if (d / H != 64 && (N > 1024 || T > 1024))
    return stlt_set_error(STLT_EINVAL, "skip-padding with head dim %lld (not 64) takes at most 1024 object slots / frames per segment (N=%lld, T=%lld)", (long long)(d / H), (long long)N, (long long)T);
  char* base = (char*)workspace;
  float* x = (float*)(base + w.x);
  float* x1 = (float*)(base + w.x1);
  float* qkv = (float*)(base + w.qkv);
  float* ctx = (float*)(base + w.ctx);
  float* tmp = (float*)(base + w.tmp);
  float* hh = (float*)(base + w.hh);
  const RaggedIndex ix = ragged_index_carve(base + w.ridx, B, T, N);
  TRY(launch_ragged_index(in->kpm_boxes, in->kpm_frames, in->lengths, B, T, N, ix, s));
  int64_t Ms = 0, Mf = 0;
  const bool host_counts = in->n_real_tokens > 0 || in->n_real_frames > 0;  // the caller knows the two row counts: no read-back, no synchronisation
  if (host_counts) {
    Ms = in->n_real_tokens; Mf = in->n_real_frames;
    TRY(launch_ragged_host_counts(ix, Ms, Mf, B * T * N, B * T, s));
  } else {
    int counts[4] = {0, 0, 0, 0};
    if (hipError_t e = hipMemcpyAsync(counts, ix.counts, sizeof(counts), hipMemcpyDeviceToHost, s); e != hipSuccess)
      return stlt_set_error((int)e, "skip-padding: count read-back: %s", hipGetErrorString(e));
    if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess)
      return stlt_set_error((int)e, "skip-padding: count read-back: %s", hipGetErrorString(e));
    if (counts[2] != 0)
      return stlt_set_error(STLT_EINVAL, "skip-padding needs collater-shaped masks: slot 0 of every real frame unmasked and frame lengths-1 real (datasets.py:247-288)");
    Ms = counts[0]; Mf = counts[1];
  }
  TRY(launch_embed(in->categories, in->boxes, in->scores, p->cat_emb, p->n_categories, p->box_w, p->box_b, p->score_w,
                   p->score_b, p->emb_ln_w, p->emb_ln_b, p->ln_eps, Ms, d, x, s, nullptr, StltDrop{0u, 1.0f, 0ull}, ix.t_orig));
  // spatial transformer: segments = frames; after it only each frame's CLS row is read (models.py:79)
  for (int64_t l = 0; l + 1 < p->n_spatial; ++l)
    TRY(encoder_layer_ragged(p->spatial[l], d, H, x, Ms, ix.t_seg_start, ix.t_seg_end, 0, STLT_K_ATTN_SPATIAL, qkv, ctx, tmp, x1, hh, x, s));
  float* cls = ctx;  // (Mf, d)
  if (p->n_spatial > 0) {
    const stlt_layer_params& lp = p->spatial[p->n_spatial - 1];
    TRY(launch_linear(x, d, lp.in_proj_w, lp.in_proj_b, qkv, 3 * d, Ms, 3 * d, d, STLT_ACT_NONE, s));
    TRY(launch_attn_ragged(qkv, ix.t_seg_start, ix.t_seg_end, 0, Ms, H, d / H, ctx, STLT_K_ATTN_SPATIAL, s));
    // qkv is dead once ctx exists: it holds the gathered rows; the layer output lands in ctx's first Mf rows
    // (ctx rows are read by the gather before anything is written back)
    TRY(encoder_tail_rows(lp, d, ctx, x, ix.f_cls_row, Mf, qkv, qkv + (size_t)Mf * d, x1, tmp, hh, qkv + (size_t)2 * Mf * d, s));
    cls = qkv + (size_t)2 * Mf * d;
  } else {
    TRY(launch_gather_rows(x, d, ix.f_cls_row, Mf, d, ctx, s));
  }
  // frames embedding on the real frames (position / frame type looked up through the frame's place in the padded batch)
  float* tbuf = x1;
  TRY(launch_frames_embed(cls, d, in->frame_types, p->pos_emb, p->type_emb, p->frames_ln_w, p->frames_ln_b, p->ln_eps, B, T,
                          d, tbuf, s, nullptr, StltDrop{0u, 1.0f, 0ull}, ix.f_orig, Mf));
  // temporal transformer: segments = clips, causal
  const int64_t tp_full = out_btd ? p->n_temporal : p->n_temporal - 1;
  for (int64_t l = 0; l < tp_full; ++l)
    TRY(encoder_layer_ragged(p->temporal[l], d, H, tbuf, Mf, ix.f_seg_start, ix.f_seg_end, 1, STLT_K_ATTN_TEMPORAL, qkv, ctx, tmp, x, hh, tbuf, s));
  if (out_btd) {
    TRY(launch_scatter_rows(tbuf, ix.f_orig, Mf, d, out_btd, B * T, s, host_counts ? ix.counts + 1 : nullptr));
    return host_counts ? launch_ragged_poison(ix, Ms, Mf, true, out_btd, B * T * d, s) : 0;
  }
  if (p->n_temporal > 0) {
    const stlt_layer_params& lp = p->temporal[p->n_temporal - 1];
    TRY(launch_linear(tbuf, d, lp.in_proj_w, lp.in_proj_b, qkv, 3 * d, Mf, 3 * d, d, STLT_ACT_NONE, s));
    TRY(launch_attn_ragged(qkv, ix.f_seg_start, ix.f_seg_end, 1, Mf, H, d / H, ctx, STLT_K_ATTN_TEMPORAL, s));
    TRY(encoder_tail_rows(lp, d, ctx, tbuf, ix.last_row, B, x, qkv, qkv + (size_t)B * d, tmp, hh, h0, s));  // qkv is dead once ctx exists
  } else {
    TRY(launch_gather_rows(tbuf, d, ix.last_row, B, d, h0, s));
  }
  // the caller's counts were taken on trust; in inference they may be upper bounds (the rows in between are dummies nobody reads).  Real counts
  // above them, or masks that break the contract: NaN rows for the head, NaN logits
  return host_counts ? launch_ragged_poison(ix, Ms, Mf, true, h0, B * d, s) : 0;
}

// exported to caf.hip (same library, C++ linkage)
int backbone_impl_public(const stlt_params* p, const stlt_inputs* in, void* workspace, size_t workspace_bytes, int flags,
                         float* out_btd, hipStream_t s) {
  TRY(check_params(p, in, false));
  if (flags & STLT_FLAG_SKIP_PADDING) {
    const WsLayout w = ws_layout(in->B, in->T, in->N, p->d, p->n_classes < 0 ? 0 : p->n_classes);
    if (!workspace || workspace_bytes < w.total)
      return stlt_set_error(STLT_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.total);
    if (!out_btd) return stlt_set_error(STLT_EINVAL, "skip-padding backbone: out_btd is null");
    StltGemmScratch gemm_scratch((char*)workspace + w.sk, STLT_GEMM_SCRATCH_BYTES);
    return forward_ragged(p, in, workspace, w, nullptr, out_btd, s);
  }
  return backbone_impl(p, in, workspace, workspace_bytes, flags, out_btd, nullptr, s);
}
size_t stlt_workspace_bytes_public(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_classes) {
  return ws_layout(B, T, N, d, n_classes < 0 ? 0 : n_classes).total;
}

extern "C" {

int stlt_backbone_forward(const stlt_params* p, const stlt_inputs* in, void* workspace, size_t workspace_bytes,
                          int flags, float* out_btd, stlt_stream_t stream) {
  TRY(check_params(p, in, false));
  if (!out_btd) return stlt_set_error(STLT_EINVAL, "stlt_backbone_forward: out_btd is null");
  return backbone_impl(p, in, workspace, workspace_bytes, flags, out_btd, nullptr, (hipStream_t)stream);
}

int stlt_forward(const stlt_params* p, const stlt_inputs* in, void* workspace, size_t workspace_bytes, int flags,
                 float* out_btd, float* logits, stlt_stream_t stream) {
  TRY(check_params(p, in, true));
  if (!logits) return stlt_set_error(STLT_EINVAL, "stlt_forward: logits is null");
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = in->B, T = in->T, N = in->N, d = p->d;
  const WsLayout w = ws_layout(B, T, N, d, p->n_classes);
  if (!workspace || workspace_bytes < w.total)
    return stlt_set_error(STLT_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.total);
  char* base = (char*)workspace;
  StltGemmScratch gemm_scratch(base + w.sk, STLT_GEMM_SCRATCH_BYTES);
  float* bb_out = out_btd ? out_btd : (float*)(base + w.x1);  // x1 doubles as the in-place temporal buffer
  float* h0 = (float*)(base + w.head);
  float* h1 = h0 + (size_t)B * d;
  float* h2 = h1 + (size_t)B * d;
  const bool last_only = (flags & STLT_FLAG_LAST_ROW_ONLY_TEMPORAL) && !out_btd && p->n_temporal > 0 && in->T > 1;
  if ((flags & STLT_FLAG_SKIP_PADDING) && !out_btd) {  // padded rows are never computed, so there is no (B,T,d) output to hand back
    TRY(forward_ragged(p, in, workspace, w, h0, nullptr, s));
  } else if (last_only) {  // the caller does not want the (B,T,d) backbone output: produce only the rows the head reads
    TRY(backbone_impl(p, in, workspace, workspace_bytes, flags, bb_out, h0, s));
  } else {
    TRY(backbone_impl(p, in, workspace, workspace_bytes, flags, bb_out, nullptr, s));
    TRY(launch_gather_last(bb_out, in->lengths, B, T, d, h0, s));                                 // models.py:189-192
  }
  TRY(launch_linear(h0, d, p->fc1_w, p->fc1_b, h1, d, B, d, d, STLT_ACT_GELU, s));                // gelu(fc1(h))
  TRY(launch_add_layernorm(h1, d, nullptr, 0, p->head_ln_w, p->head_ln_b, p->ln_eps, B, d, h2, d, s));
  TRY(launch_linear(h2, d, p->fc2_w, p->fc2_b, logits, p->n_classes, B, p->n_classes, d, STLT_ACT_NONE, s));
  return 0;
}

}  // extern "C"
