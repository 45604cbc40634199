// Training step of the STLT path in native code: the forward that records a tape, and the reverse sweep that
// autograd would run for the reference's `loss.backward()` (src/train.py:125-127).  Both are fixed launch
// sequences on the caller's stream (no allocation, no synchronisation).
//
// Tape (fp32, rows padded to a multiple of 32 and zero beyond the logical row count — the caller allocates it
// zero-filled and the kernels never write the padding, which is what lets dW = dYᵀ·X run with a rounded-up
// contraction length):
//   s_embed (tok,d) | per spatial layer: x, qkv(3d), ctx, a, x1, u(4d), h(4d), f | x_sp_out | s_frames (BT,d)
//   | per temporal layer: same 8 buffers on BT rows | x_tp_out | head: h0, u0, z1, z2 (B rows)
//   layer math:  qkv = x·Winᵀ+b ; ctx = attn(qkv) ; a = ctx·Woᵀ+bo ; x1 = LN1(x+a) ; u = x1·W1ᵀ+b1 ; h = gelu(u) ;
//                f = h·W2ᵀ+b2 ; y = LN2(x1+f)   (y is the next layer's x)
#include <cstdlib>
#include <mutex>
#include "ctx.h"

namespace {

inline int64_t up32(int64_t v) { return (v + 31) / 32 * 32; }
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

#define TRY(expr) do { int _e = (expr); if (_e) return _e; } while (0)

struct LayerTape { float *x, *qkv, *ctx, *a, *x1, *u, *h, *f; };

struct Tape {
  int64_t tokp, btp, bp;
  float* s_embed;
  LayerTape sp[64];
  float* sp_out;
  float* s_frames;
  LayerTape tp[64];
  float* tp_out;
  float *h0, *u0, *z1, *z2;
  float* sk;  // stream-K partial tiles (not part of the record: scratch of the forward's GEMM launches)
  char* ridx;  // STLT_FLAG_SKIP_PADDING: the ragged index built by the forward, read again by the backward
  size_t bytes;
};

// carve the tape; base may be null to just size it
static Tape tape_layout(char* base, int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_sp, int64_t n_tp) {
  Tape t;
  t.tokp = up32(B * T * N); t.btp = up32(B * T); t.bp = up32(B);
  size_t off = 0;
  auto take = [&](int64_t rows, int64_t width) { float* p = base ? (float*)(base + off) : nullptr; off = align256(off + (size_t)rows * width * sizeof(float)); return p; };
  t.s_embed = take(t.tokp, d);
  auto take_layer = [&](LayerTape& l, int64_t rows) {
    l.x = take(rows, d); l.qkv = take(rows, 3 * d); l.ctx = take(rows, d); l.a = take(rows, d);
    l.x1 = take(rows, d); l.u = take(rows, 4 * d); l.h = take(rows, 4 * d); l.f = take(rows, d);
  };
  for (int64_t l = 0; l < n_sp; ++l) take_layer(t.sp[l], t.tokp);
  t.sp_out = take(t.tokp, d);
  t.s_frames = take(t.btp, d);
  for (int64_t l = 0; l < n_tp; ++l) take_layer(t.tp[l], t.btp);
  t.tp_out = take(t.btp, d);
  t.h0 = take(t.bp, d); t.u0 = take(t.bp, d); t.z1 = take(t.bp, d); t.z2 = take(t.bp, d);
  t.sk = take((int64_t)(STLT_GEMM_SCRATCH_BYTES / sizeof(float)), 1);
  t.ridx = (char*)take((int64_t)((ragged_index_bytes(B, T, N) + 3) / 4), 1);
  t.bytes = off;
  return t;
}

// backward scratch: gradient buffers for the spatial phase (tok rows) and, separately, the temporal phase (BT
// rows) so that each one's row padding stays zero; split-K slabs; reduction scratch.
// The output gradients of a layer's four Linears (df, du, da, dqkv) are the operands of its weight-gradient products, which run on
// a second stream beside the NEXT layer's dX chain: two sets per tower, used by alternate layers (GradBufs).
struct GradBufs { float *B, *D, *E, *Q, *H; };
constexpr int DW_EXTRA_SETS = 6;            // + the tower's own set + the second set = 8 layers per grouped launch
constexpr int64_t DW_DEFER_MAX_ROWS = 4096;  // beyond this the products are long enough by themselves (weight_grad_all) and the sets large
struct Scratch {
  float *sA, *sB, *sC, *sD, *sE, *sQKV, *sH;  // spatial: (tokp,d) x5, (tokp,3d), (tokp,4d)
  float *tA, *tB, *tC, *tD, *tE, *tQKV, *tH;  // temporal
  GradBufs s2, t2;                  // the second sets
  GradBufs tx[DW_EXTRA_SETS];       // temporal tower, few rows (<= DW_DEFER_MAX_ROWS): more sets, so that the weight gradients of up to
  int n_tx;                         //   eight layers run as ONE grouped launch (32 products) instead of one launch per layer
  float* sk2;                       // stream-K partial tiles of the second stream's launches
  float *hA, *hB;                   // head: (bp,d) x2
  float* slabs;
  float* red;
  float* sk;
  size_t slab_floats, bytes;
  float* red_pool;                  // chunks of `red_floats` for the sweep's partial rows while their reductions are deferred (StltReduceDefer)
  size_t red_floats, red_pool_floats;
  StltReduceDefer* defer = nullptr;
};
// the partial-row scratch of the next producer: a fresh chunk of the pool while reductions are deferred, else the one shared buffer
static float* RED(const Scratch& sc) {
  int err = 0;
  return stlt_reduce_defer_chunk(sc.defer, sc.red_floats, sc.red, &err);
}

constexpr int MAX_SPLIT = 32;
constexpr int AB_MAX_ROWS = 256;  // attention backward: longest sequence (backward.hip: 64 in LDS, up to 256 streamed)

static bool dw_side_wanted();
static bool defer_wanted() { static const bool on = [] { const char* e = getenv("STLT_TRAIN_DEFER_REDUCE"); return e ? atoi(e) != 0 : true; }(); return on; }

// The optional parts follow the switches the sweep itself reads, so that callers of stlt_train_scratch_bytes do not pay for what a
// configuration never touches: the spatial tower's second operand set (10 tokp d floats: +7 GB at 1024 clips of cfg2) only while the
// side stream is wanted, the reduction pool (<= 256 MB) only while the partial-row reductions are deferred.  They sit BEHIND the fixed
// part, so toggling a switch between two calls moves no fixed buffer; the size is asked again by every stlt_train_backward caller.
static Scratch scratch_layout(char* base, int64_t B, int64_t T, int64_t N, int64_t d, int64_t C) {
  Scratch s;
  const int64_t tokp = up32(B * T * N), btp = up32(B * T), bp = up32(B);
  size_t off = 0;
  auto take = [&](int64_t floats) { float* p = base ? (float*)(base + off) : nullptr; off = align256(off + (size_t)floats * sizeof(float)); return p; };
  s.sA = take(tokp * d); s.sB = take(tokp * d); s.sC = take(tokp * d); s.sD = take(tokp * d); s.sE = take(tokp * d); s.sQKV = take(tokp * 3 * d); s.sH = take(tokp * 4 * d);
  s.tA = take(btp * d); s.tB = take(btp * d); s.tC = take(btp * d); s.tD = take(btp * d); s.tE = take(btp * d); s.tQKV = take(btp * 3 * d); s.tH = take(btp * 4 * d);
  s.hA = take(bp * d); s.hB = take(bp * d);
  s.slab_floats = (size_t)MAX_SPLIT * 4 * d * d;
  s.slabs = take((int64_t)s.slab_floats);
  int64_t red = ln_bwd_scratch_floats(d);
  if (512 * 4 * d > red) red = 512 * 4 * d;  // gelu_bwd column-sum partials (512 x 4d); attn_bwd needs 256 x 3d
  if ((tokp + 255) / 256 * 16 * 4 * d > red) red = (tokp + 255) / 256 * 16 * 4 * d;  // fused GELU backward: 16 partial rows per 256-row tile row
  if (16 * (T + 5) * d > red) red = 16 * (T + 5) * d;  // frames-embedding parameter partials
  const int64_t eb = embed_bwd_scratch_floats(B * T * N, C, d);
  if (eb > red) red = eb;
  s.red = take(red);
  s.red_floats = (size_t)red;
  s.sk = take((int64_t)(STLT_GEMM_SCRATCH_BYTES / sizeof(float)));
  s.t2 = GradBufs{take(btp * d), take(btp * d), take(btp * d), take(btp * 3 * d), take(btp * 4 * d)};
  s.n_tx = btp <= DW_DEFER_MAX_ROWS ? DW_EXTRA_SETS : 0;
  for (int i = 0; i < s.n_tx; ++i) s.tx[i] = GradBufs{take(btp * d), take(btp * d), take(btp * d), take(btp * 3 * d), take(btp * 4 * d)};
  // ---- optional parts
  s.red_pool_floats = (size_t)red * 24 < ((size_t)64 << 20) ? (size_t)red * 24 : ((size_t)64 << 20);  // <= 256 MB
  if (s.red_pool_floats < (size_t)red || !defer_wanted()) s.red_pool_floats = 0;  // a chunk would not fit, or no deferral
  s.red_pool = s.red_pool_floats ? take((int64_t)s.red_pool_floats) : nullptr;
  const bool side = dw_side_wanted();
  s.s2 = side ? GradBufs{take(tokp * d), take(tokp * d), take(tokp * d), take(tokp * 3 * d), take(tokp * 4 * d)} : GradBufs{nullptr, nullptr, nullptr, nullptr, nullptr};
  s.sk2 = side ? take((int64_t)(STLT_GEMM_SCRATCH_BYTES / sizeof(float))) : nullptr;
  s.bytes = off;
  return s;
}

// ---- second stream for the weight-gradient products --------------------------------------------------------------------------
// A layer's four weight gradients are off the dX chain (nothing downstream reads them before the optimiser).  They are enqueued
// on a per-device side stream (lower priority: the chain is the critical path) behind an event the chain records, with their own
// stream-K scratch; the chain waits for them only when it is about to reuse their operand buffers (two layers later) and at the
// end of the call.  The products fill the chip while the chain runs its row-wise kernels, fix-ups and under-filled launches.
// STLT_TRAIN_DW_STREAM=0 keeps everything on the caller's stream (A/B runs); STLT_TRAIN_DW_WG / STLT_TRAIN_DX_WG cap the grids of
// the side products / the chain's dX products (0 = uncapped) so that both persistent kernels can be resident at once.
struct DwSide {
  StltSideDevice* dev = nullptr;  // != nullptr: this sweep holds dev->busy (released by DwSideHold)
  hipStream_t s = nullptr;
  hipEvent_t chain[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
  bool pending[2] = {false, false};
  bool on = false;
  float* sk = nullptr;
  int flush_no = 0;
};
static int g_dw_side_wanted = -1;  // -1: not read yet (STLT_TRAIN_DW_STREAM, default on); stlt_set_train_side_stream overrides
static bool dw_side_wanted() {
  if (g_dw_side_wanted < 0) { const char* e = getenv("STLT_TRAIN_DW_STREAM"); g_dw_side_wanted = e ? (atoi(e) != 0) : 1; }
  return g_dw_side_wanted != 0;
}
static int dw_side_wg_cap() { static const int n = [] { const char* e = getenv("STLT_TRAIN_DW_WG"); return e ? atoi(e) : 0; }(); return n; }
static int dx_chain_wg_cap() { static const int n = [] { const char* e = getenv("STLT_TRAIN_DX_WG"); return e ? atoi(e) : 0; }(); return n; }

// The side stream and its events belong to the call's context (ctx.h: one set per context and device, created at the first sweep that wants it,
// destroyed with the context); a call that names no context keeps every launch on the caller's stream.
static DwSide dw_side_open(const Scratch& sc, stlt_ctx* ctx) {
  DwSide sd;
  if (!ctx || !dw_side_wanted() || !sc.sk2) return sd;
  StltSideDevice& dv = ctx->side[stlt_current_device() & (STLT_MAX_DEVICES - 1)];
  dv.busy.lock();  // also serialises the one creation per device, whichever thread's sweep comes first
  sd.dev = &dv;
  if (!dv.tried) {
    dv.tried = true;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent
    bool ok = hipStreamCreateWithPriority(&dv.s, hipStreamNonBlocking, lo) == hipSuccess;
    for (int i = 0; ok && i < 4; ++i) ok = hipEventCreateWithFlags(&dv.ev[i], hipEventDisableTiming) == hipSuccess;
    dv.ok = ok;
    (void)hipGetLastError();
  }
  if (!dv.ok) return sd;
  sd.s = dv.s;
  sd.chain[0] = dv.ev[0]; sd.chain[1] = dv.ev[1]; sd.done[0] = dv.ev[2]; sd.done[1] = dv.ev[3];
  sd.sk = sc.sk2;
  sd.on = true;
  return sd;
}
// the chain is about to write the operand buffers of set `par`: the side products that still read them must have finished
static int dw_side_wait(DwSide* sd, int par, hipStream_t s) {
  if (sd && sd->on && sd->pending[par]) {
    if (hipError_t e = hipStreamWaitEvent(s, sd->done[par], 0); e != hipSuccess) return stlt_set_error((int)e, "train_backward: %s", hipGetErrorString(e));
    sd->pending[par] = false;
  }
  return 0;
}
static int dw_side_join(DwSide* sd, hipStream_t s) {
  TRY(dw_side_wait(sd, 0, s));
  return dw_side_wait(sd, 1, s);
}
// Owner of a sweep's DwSide: on EVERY exit path of the call (error returns included) the caller's stream is made to wait for the
// side products still in flight — they read caller-owned tape and scratch, which torch's allocator recycles by the caller's stream
// alone, and a captured graph must see the fork joined — and the per-device stream / event set is handed back.
struct DwSideHold {
  DwSide* sd; hipStream_t s;
  DwSideHold(DwSide* x, hipStream_t st) : sd(x), s(st) {}
  ~DwSideHold() {
    (void)dw_side_join(sd, s);
    if (sd->dev) { sd->dev->busy.unlock(); sd->dev = nullptr; }
  }
  DwSideHold(const DwSideHold&) = delete;
  DwSideHold& operator=(const DwSideHold&) = delete;
};

// The weight-gradient launches of one tower.  Layers take the operand-buffer sets in turn; the products of `group_layers` consecutive
// layers are collected and flushed as one grouped launch — on the side stream when it is on (behind an event of the chain), else on the
// caller's stream.  A set is rewritten only after the flush that read it has finished.
struct DwQueue {
  DwSide* side = nullptr;
  GradBufs sets[2 + DW_EXTRA_SETS];
  int n_sets = 1, group_layers = 1;
  StltWeightGradItem items[STLT_GEMM_GROUP_MAX];
  int n_items = 0, layers_in_chunk = 0, layer_no = 0;
  int set_flush[2 + DW_EXTRA_SETS];  // parity of the side-stream flush still reading the set, -1 = none
  int chunk_sets[2 + DW_EXTRA_SETS];  // sets of the layers collected since the last flush
  DwQueue() { for (int& f : set_flush) f = -1; }
};
static int dwq_wait_parity(DwQueue& q, int par, hipStream_t s) {
  TRY(dw_side_wait(q.side, par, s));
  for (int k = 0; k < q.n_sets; ++k) if (q.set_flush[k] == par) q.set_flush[k] = -1;
  return 0;
}
static int dwq_begin_layer(DwQueue& q, hipStream_t s, GradBufs& out, int& set_idx) {
  set_idx = q.layer_no++ % q.n_sets;
  if (q.set_flush[set_idx] >= 0) TRY(dwq_wait_parity(q, q.set_flush[set_idx], s));
  out = q.sets[set_idx];
  return 0;
}
static int dwq_flush(DwQueue& q, const Scratch& sc, hipStream_t s);

static int n_cu_cached() { return stlt_device_cus(); }

// g_w (n_out, k_in) += dYᵀ·X with dY (Mp, n_out), X (Mp, k_in).  A weight matrix is only 18-72 output tiles, so the
// launch runs as stream-K over the token contraction (every CU gets an equal share of k-steps; the fix-up adds the
// partial tiles to g_w in a fixed order).  Without lent scratch it falls back to explicit split-K slabs.
static int weight_grad(const float* dy, int64_t n_out, const float* x, int64_t k_in, int64_t Mp, float* g_w, const Scratch& sc,
                       hipStream_t s) {
  if (!g_w) return 0;
  if (sc.sk) return launch_gemm(1, 1, dy, n_out, x, k_in, nullptr, g_w, k_in, g_w, k_in, 0, n_out, k_in, Mp, 1, STLT_ACT_NONE, s);
  const int64_t tiles = ((n_out + 255) / 256) * ((k_in + 127) / 128);
  const int64_t steps = Mp / 32;
  int64_t want = (2 * n_cu_cached() + tiles - 1) / tiles;
  if (want > MAX_SPLIT) want = MAX_SPLIT;
  int split = 1;
  for (int64_t c = 1; c <= want; ++c) if (steps % c == 0) split = (int)c;
  TRY(launch_gemm(1, 1, dy, n_out, x, k_in, nullptr, nullptr, 0, sc.slabs, k_in, n_out * k_in, n_out, k_in, Mp, split,
                  STLT_ACT_NONE, s));
  return launch_reduce_slabs(sc.slabs, n_out * k_in, split, g_w, n_out * k_in, 1, s);
}

// C (rows, k_in) = dY (rows, n_out)·W (n_out, k_in) (+ R): the input gradient of a Linear.  With the opt-in split-bf16 products
// switched on and a launch they take, W is transposed into the (then unused) split-K slab buffer and the product runs as the
// NT form the split-bf16 kernel has (gemm_bf16x3.hip: launch_input_grad_bf16x3); otherwise the f32-MFMA NN kernel.
static int dx_product(const float* dy, int64_t ld_dy, const float* w, int64_t n_out, int64_t k_in, const float* r, int64_t ldr, float* c,
                      int64_t ldc, int64_t rows, const Scratch& sc, hipStream_t s) {
  if (sc.sk && (size_t)(n_out * k_in) <= sc.slab_floats) {
    bool taken = false;
    TRY(launch_input_grad_bf16x3(dy, ld_dy, w, n_out, k_in, r, ldr, c, ldc, rows, sc.slabs, s, &taken));
    if (taken) return 0;
  }
  bool small = false;  // under-filled launches (few rows): whole small tiles, W read as it lies (gemm16.hip)
  TRY(launch_input_grad_gemm16(dy, ld_dy, w, n_out, k_in, r, ldr, c, ldc, rows, s, &small));
  if (small) return 0;
  return launch_gemm(0, 1, dy, ld_dy, w, k_in, nullptr, r, ldr, c, ldc, 0, rows, k_in, n_out, 1, STLT_ACT_NONE, s);
}

// The weight gradients of a layer as ONE grouped stream-K launch (gemm.hip: launch_weight_grad_group) when stream-K
// scratch is lent, else product by product.  Every CU gets an equal share of the four products' k-steps: one pipeline
// fill, at most two partial tiles per workgroup and one fix-up for the layer instead of four of each.
static int weight_grad_all(const StltWeightGradItem* items, int n, const Scratch& sc, hipStream_t s) {
  static const bool grouped = [] { const char* e = getenv("STLT_GEMM_GROUP_DW"); return e ? atoi(e) != 0 : true; }();  // A/B knob
  // Grouping pays where the products are launch-bound (the temporal tower at 64 clips: 54 k-steps per workgroup for all four,
  // 322 -> 266 us); with long contractions (the spatial tower: 378 k-steps per workgroup) the four separate launches are
  // slightly faster — three interleaved A/B pairs of the 64-clip step: 27.17 ms grouped everywhere, 27.07 ms with this limit
  static const int64_t max_rows = [] { const char* e = getenv("STLT_GEMM_GROUP_DW_MAXROWS"); return e ? (int64_t)atoll(e) : (int64_t)4096; }();
  int64_t rows = 0;
  for (int i = 0; i < n; ++i) if (items[i].rows > rows) rows = items[i].rows;
  if (grouped && rows <= max_rows && sc.sk && stlt_gemm_has_scratch()) return launch_weight_grad_group(items, n, s);
  for (int i = 0; i < n; ++i) TRY(weight_grad(items[i].dy, items[i].n_out, items[i].x, items[i].k_in, items[i].rows, items[i].g_w, sc, s));
  return 0;
}

// dh = df·W2 followed by du = drop(dh) ∘ gelu'(u) and lin1_b += column sums of du.  Fused (default): the activation
// derivative, the dropout mask and the column sums ride in the product's epilogue / fix-up (gemm.hip: STLT_ACT_GELU_BWD) and
// one reduction finishes the bias gradient; STLT_FUSE_GELU_BWD=0 keeps the stand-alone pass behind the product (A/B runs).
static int ffn_hidden_backward(const float* df, const float* lin2_w, const float* u, float* du, int64_t rows, int64_t d, float* g_lin1_b,
                               const Scratch& sc, StltDrop dr, uint32_t site, const int* drop_rows, hipStream_t s);
}  // namespace
// The same for the block-level training calls (blocks.hip): du = drop(df·W2) ∘ gelu'(u) in the dX product's epilogue, lin1_b += colsum(du)
// through `cs_part` (>= ceil(rows / 256) * 16 * 4d floats; with deferred reductions a chunk of the caller's pool).  Needs lent stream-K
// scratch on the calling thread like every under-filled product; *taken = false when the fused form does not apply.
int stlt_ffn_hidden_backward_fused(const float* df, const float* lin2_w, const float* u, float* du, int64_t rows, int64_t d, float* g_lin1_b,
                                   float* cs_part, StltDrop dr, uint32_t site, hipStream_t s, bool* taken) {
  static const bool fused = [] { const char* e = getenv("STLT_FUSE_GELU_BWD"); return e ? atoi(e) != 0 : true; }();
  *taken = false;
  if (!fused || !g_lin1_b || !cs_part || d % 32 != 0 || stlt_split_bf16_takes(rows, 4 * d, d, d, d)) return 0;
  *taken = true;
  const StltGemmEpi epi{dr, site, nullptr, cs_part};
  bool small = false;  // under-filled launches (the fusion models' 2048 / 2112-row blocks): whole small tiles with the same epilogue (gemm16.hip)
  if (int e = launch_input_grad_gemm16(df, d, lin2_w, d, 4 * d, u, 4 * d, du, 4 * d, rows, s, &small, 0, &epi)) return e;
  if (!small) { if (int e = launch_gemm(0, 1, df, d, lin2_w, 4 * d, nullptr, u, 4 * d, du, 4 * d, 0, rows, 4 * d, d, 1, STLT_ACT_GELU_BWD, s, &epi)) return e; }
  return launch_reduce_slabs(cs_part, 4 * d, (int)((rows + 255) / 256 * 16), g_lin1_b, 4 * d, 1, s);
}
namespace {
static int ffn_hidden_backward(const float* df, const float* lin2_w, const float* u, float* du, int64_t rows, int64_t d, float* g_lin1_b,
                               const Scratch& sc, StltDrop dr, uint32_t site, const int* drop_rows, hipStream_t s) {
  static const bool fused = [] { const char* e = getenv("STLT_FUSE_GELU_BWD"); return e ? atoi(e) != 0 : true; }();
  if (fused && g_lin1_b && d % 32 == 0 && !(sc.sk && stlt_split_bf16_takes(rows, 4 * d, d, d, d))) {  // (the split-bf16 product has no GELU-backward epilogue)
    float* cs = RED(sc);
    const StltGemmEpi epi{dr, site, drop_rows, cs};
    bool small = false;  // the temporal tower at the reference's default batch (2048 rows): whole small tiles with the same epilogue (gemm16.hip)
    TRY(launch_input_grad_gemm16(df, d, lin2_w, d, 4 * d, u, 4 * d, du, 4 * d, rows, s, &small, 0, &epi));
    if (!small) TRY(launch_gemm(0, 1, df, d, lin2_w, 4 * d, nullptr, u, 4 * d, du, 4 * d, 0, rows, 4 * d, d, 1, STLT_ACT_GELU_BWD, s, &epi));
    return launch_reduce_slabs(cs, 4 * d, (int)((rows + 255) / 256 * 16), g_lin1_b, 4 * d, 1, s);
  }
  TRY(dx_product(df, d, lin2_w, d, 4 * d, nullptr, 0, du, 4 * d, rows, sc, s));  // dh
  if (g_lin1_b) return launch_gelu_bwd_colsum(du, u, du, rows, 4 * d, g_lin1_b, RED(sc), s, dr, site, drop_rows);  // du; lin1_b += colsum(du)
  return launch_gelu_bwd(du, u, du, rows * 4 * d, s, dr, site, drop_rows, 4 * d);
}

// backward of one encoder layer.  dy: gradient wrt the layer output (M,d) in bufA; on return bufA holds the gradient
// wrt the layer input.  bufB / bufC / bufD / bufE (M,d), bufQ (M,3d), bufH (M,4d) are scratch with zero row padding.
// The four output gradients of the layer's Linears (df, du, da, dqkv) stay alive until the end of the layer, where their
// weight gradients run as one grouped launch: df in bufD (dropout on: the gradient wrt the un-dropped branch output,
// ds * mask / (1-p)) or bufB (dropout off: the residual-path gradient is the branch gradient), du in bufH, da in bufE,
// dqkv in bufQ.
static int layer_backward(const stlt_layer_params& lp, const stlt_layer_params* g, const LayerTape& t, int64_t d, int64_t H,
                          int64_t M, int64_t Mp, int64_t S, int64_t L, const uint8_t* kpm, int causal, float* bufA, float* bufB,
                          float* bufC, float* bufD, float* bufE, float* bufQ, float* bufH, const Scratch& sc, StltDrop dr, uint32_t site0,
                          hipStream_t s, const AttnBwdRagged* rg = nullptr, DwQueue* q = nullptr) {
  auto G = [&](const float* stlt_layer_params::*m) -> float* { return g ? const_cast<float*>(g->*m) : nullptr; };
  // with a queue, the layer takes the next set of operand buffers (waiting for the flush that still reads it, if any) and leaves its
  // four weight-gradient products with the queue
  int set_idx = 0;
  if (q) {
    GradBufs gb;
    TRY(dwq_begin_layer(*q, s, gb, set_idx));
    bufB = gb.B; bufD = gb.D; bufE = gb.E; bufQ = gb.Q; bufH = gb.H;
  }
  StltGemmWgCap chain_cap(q && q->side && q->side->on ? dx_chain_wg_cap() : 0);
  float* df = dr.thr ? bufD : bufB;   // gradient wrt f (after the dropout mask)
  float* ds1 = dr.thr ? bufB : bufE;  // residual-path gradient behind norm1 (bufB's ds2 is dead by then when dropout is on)
  float* da = bufE;                   // gradient wrt a
  // y = LN2(x1 + drop(f))
  TRY(launch_ln_bwd(bufA, d, t.x1, d, t.f, d, lp.norm2_w, 1e-5f, M, d, bufB, d, G(&stlt_layer_params::norm2_w),
                    G(&stlt_layer_params::norm2_b), RED(sc), s, dr, site0 + 3, bufD, 0,
                    G(&stlt_layer_params::lin2_b)));                                               // bufB = ds2, df; lin2_b += colsum(df)
  // f = h·W2ᵀ + b2
  // h = drop(gelu(u)): bufH = du = drop(df·W2) ∘ gelu'(u); lin1_b += colsum(du)
  TRY(ffn_hidden_backward(df, lp.lin2_w, t.u, bufH, M, d, G(&stlt_layer_params::lin1_b), sc, dr, site0 + 2, nullptr, s));
  // u = x1·W1ᵀ + b1
  TRY(dx_product(bufH, 4 * d, lp.lin1_w, 4 * d, d, bufB, d, bufC, d, M, sc, s));  // bufC = dx1 = du·W1 + ds2
  // x1 = LN1(x + drop(a))
  TRY(launch_ln_bwd(bufC, d, t.x, d, t.a, d, lp.norm1_w, 1e-5f, M, d, ds1, d, G(&stlt_layer_params::norm1_w),
                    G(&stlt_layer_params::norm1_b), RED(sc), s, dr, site0 + 1, bufE, 0,
                    G(&stlt_layer_params::out_proj_b)));                                           // ds1, da; out_proj_b += colsum(da)
  // a = ctx·Woᵀ + bo
  TRY(dx_product(da, d, lp.out_proj_w, d, d, nullptr, 0, bufC, d, M, sc, s));  // bufC = dctx
  // ctx = attention(qkv) with dropout on the probabilities
  TRY(launch_attn_bwd(t.qkv, bufC, kpm, causal, S, L, H, d / H, bufQ, s, dr, site0, G(&stlt_layer_params::in_proj_b), RED(sc), rg));  // bufQ = dqkv; in_proj_b += colsum(dqkv)
  // qkv = x·Winᵀ + bin
  TRY(dx_product(bufQ, 3 * d, lp.in_proj_w, 3 * d, d, ds1, d, bufA, d, M, sc, s));  // bufA = dx = dqkv·Win + ds1
  // the four weight gradients (off the dX chain): one grouped launch
  const StltWeightGradItem items[4] = {{df, d, t.h, 4 * d, Mp, G(&stlt_layer_params::lin2_w)},
                                       {bufH, 4 * d, t.x1, d, Mp, G(&stlt_layer_params::lin1_w)},
                                       {da, d, t.ctx, d, Mp, G(&stlt_layer_params::out_proj_w)},
                                       {bufQ, 3 * d, t.x, d, Mp, G(&stlt_layer_params::in_proj_w)}};
  if (q) {
    for (int i = 0; i < 4; ++i) q->items[q->n_items++] = items[i];
    q->chunk_sets[q->layers_in_chunk++] = set_idx;
    if (q->layers_in_chunk >= q->group_layers || q->n_items + 4 > STLT_GEMM_GROUP_MAX) return dwq_flush(*q, sc, s);
    return 0;
  }
  return weight_grad_all(items, 4, sc, s);
}

// launch what the queue has collected: one grouped launch (weight_grad_all groups products of <= 4096 rows) on the side stream behind
// the chain's event, or on the caller's stream
static int dwq_flush(DwQueue& q, const Scratch& sc, hipStream_t s) {
  if (q.n_items == 0) return 0;
  DwSide* side = q.side;
  if (side && side->on) {
    const int par = side->flush_no++ & 1;
    TRY(dwq_wait_parity(q, par, s));  // the events of this parity are re-recorded below: nothing may still be waiting on them
    if (hipError_t e = hipEventRecord(side->chain[par], s); e != hipSuccess) return stlt_set_error((int)e, "train_backward: %s", hipGetErrorString(e));
    if (hipError_t e = hipStreamWaitEvent(side->s, side->chain[par], 0); e != hipSuccess) return stlt_set_error((int)e, "train_backward: %s", hipGetErrorString(e));
    int rc = 0;
    {
      StltGemmScratch lend(side->sk, STLT_GEMM_SCRATCH_BYTES);
      StltGemmWgCap cap(dw_side_wg_cap());
      rc = weight_grad_all(q.items, q.n_items, sc, side->s);
    }
    // also after a failed launch: whatever did get enqueued on the side stream is joined by the call's exit path (DwSideHold)
    if (hipError_t e = hipEventRecord(side->done[par], side->s); e != hipSuccess) return stlt_set_error((int)e, "train_backward: %s", hipGetErrorString(e));
    side->pending[par] = true;
    TRY(rc);
    for (int i = 0; i < q.layers_in_chunk; ++i) q.set_flush[q.chunk_sets[i]] = par;
  } else {
    TRY(weight_grad_all(q.items, q.n_items, sc, s));
  }
  q.n_items = 0;
  q.layers_in_chunk = 0;
  return 0;
}

// in-projection + attention core of a training forward: the packed projections stay in the tape (t.qkv) for the reverse sweep.
// Padded layout, sequences of <= 64 tokens: one fused launch (mhsa.hip, TRAIN build: dropout of the probabilities from the
// counter mask, q / k / v written from the accumulators); ragged layout and other shapes: product + attention core.
static int qkv_attention_train(const stlt_layer_params& lp, int64_t d, int64_t H, const LayerTape& t, int64_t M, int64_t S, int64_t L,
                               const uint8_t* kpm, int causal, int kid, StltDrop dr, uint32_t site0, hipStream_t s, const int* seg_start,
                               const int* seg_end) {
  if (!seg_start && stlt_fused_mhsa_on(causal) && stlt_mhsa_fused_pays(S, L, H, d, causal) && !stlt_split_bf16_takes(M, 3 * d, d, d, d))
    return launch_mhsa_fused(t.x, lp.in_proj_w, lp.in_proj_b, kpm, S, L, H, d, t.ctx, s, causal, t.qkv, dr, site0);
  TRY(launch_linear(t.x, d, lp.in_proj_w, lp.in_proj_b, t.qkv, 3 * d, M, 3 * d, d, STLT_ACT_NONE, s));
  if (seg_start) return launch_attn_ragged(t.qkv, seg_start, seg_end, causal, M, H, d / H, t.ctx, kid, s, dr, site0);
  return launch_attn(t.qkv, kpm, causal, S, L, H, d / H, t.ctx, kid, s, dr, site0);
}

static int layer_forward(const stlt_layer_params& lp, int64_t d, int64_t H, const LayerTape& t, int64_t M, int64_t S, int64_t L,
                         const uint8_t* kpm, int causal, int kid, float* y, StltDrop dr, uint32_t site0, hipStream_t s,
                         const int* seg_start = nullptr, const int* seg_end = nullptr) {
  TRY(qkv_attention_train(lp, d, H, t, M, S, L, kpm, causal, kid, dr, site0, s, seg_start, seg_end));
  TRY(launch_linear(t.ctx, d, lp.out_proj_w, lp.out_proj_b, t.a, d, M, d, d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(t.a, d, t.x, d, lp.norm1_w, lp.norm1_b, 1e-5f, M, d, t.x1, d, s, dr, site0 + 1));
  TRY(launch_linear_gelu_keep(t.x1, d, lp.lin1_w, lp.lin1_b, t.u, t.h, M, 4 * d, d, dr, site0 + 2, nullptr, s));
  TRY(launch_linear(t.h, 4 * d, lp.lin2_w, lp.lin2_b, t.f, d, M, d, 4 * d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(t.f, d, t.x1, d, lp.norm2_w, lp.norm2_b, 1e-5f, M, d, y, d, s, dr, site0 + 3));
  return 0;
}

// Last layer of a tower when only n of its M output rows are read afterwards (the CLS row of every frame after
// the spatial tower, models.py:79; frame lengths-1 of every clip after the temporal tower, models.py:189-192): the
// in-projection and the attention run on all rows (every row is a key / value), out-proj, norms and FFN on the picked
// rows only.  Tape: x, qkv, ctx hold M rows; a, x1, u, h, f hold n rows; the gathered ctx / x rows are parked in the
// unused upper part of a / x1 (rows n..2n-1, hence the 2n <= M condition at the call sites) and re-gathered by the
// reverse sweep.  Dropout masks keep the indices of the rows' original positions.
static int layer_forward_tail(const stlt_layer_params& lp, int64_t d, int64_t H, const LayerTape& t, int64_t M, int64_t S, int64_t L,
                              const uint8_t* kpm, int causal, int kid, const int* seg_start, const int* seg_end, const int* rows,
                              int64_t n, float* y, StltDrop dr, uint32_t site0, hipStream_t s) {
  TRY(qkv_attention_train(lp, d, H, t, M, S, L, kpm, causal, kid, dr, site0, s, seg_start, seg_end));
  float* g_ctx = t.a + n * d;
  float* g_x = t.x1 + n * d;
  TRY(launch_gather_rows(t.ctx, d, rows, n, d, g_ctx, s));
  TRY(launch_gather_rows(t.x, d, rows, n, d, g_x, s));
  TRY(launch_linear(g_ctx, d, lp.out_proj_w, lp.out_proj_b, t.a, d, n, d, d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(t.a, d, g_x, d, lp.norm1_w, lp.norm1_b, 1e-5f, n, d, t.x1, d, s, dr, site0 + 1, rows));
  TRY(launch_linear_gelu_keep(t.x1, d, lp.lin1_w, lp.lin1_b, t.u, t.h, n, 4 * d, d, dr, site0 + 2, rows, s));
  TRY(launch_linear(t.h, 4 * d, lp.lin2_w, lp.lin2_b, t.f, d, n, d, 4 * d, STLT_ACT_NONE, s));
  TRY(launch_add_layernorm(t.f, d, t.x1, d, lp.norm2_w, lp.norm2_b, 1e-5f, n, d, y, d, s, dr, site0 + 3, rows));
  return 0;
}

static int zero_rows(float* buf, int64_t width, int64_t r0, int64_t r1, hipStream_t s) {
  if (r1 <= r0) return 0;
  if (hipError_t e = hipMemsetAsync(buf + r0 * width, 0, (size_t)(r1 - r0) * width * sizeof(float), s); e != hipSuccess)
    return stlt_set_error((int)e, "train_backward: memset: %s", hipGetErrorString(e));
  return 0;
}

// Reverse of layer_forward_tail.  dy: gradient wrt the n output rows; on return bufA holds the gradient wrt all M
// input rows.  Scratch roles as in layer_backward; the gradient-side operands of the weight-gradient products are
// zeroed between n and its round-up to 32 first (those rows belong to other layers' data in the shared buffers).
static int layer_backward_tail(const stlt_layer_params& lp, const stlt_layer_params* g, const LayerTape& t, int64_t d, int64_t H,
                               int64_t M, int64_t Mp, int64_t S, int64_t L, const uint8_t* kpm, int causal, const int* rows, int64_t n,
                               const float* dy, float* bufA, float* bufB, float* bufC, float* bufD, float* bufE, float* bufQ, float* bufH,
                               const Scratch& sc, StltDrop dr, uint32_t site0, hipStream_t s, const AttnBwdRagged* rg) {
  auto G = [&](const float* stlt_layer_params::*m) -> float* { return g ? const_cast<float*>(g->*m) : nullptr; };
  const int64_t np = up32(n);
  float* df = dr.thr ? bufD : bufB;
  float* ds1 = dr.thr ? bufB : bufE;
  float* da = bufE;
  TRY(zero_rows(bufB, d, n, np, s));
  TRY(zero_rows(bufD, d, n, np, s));
  TRY(zero_rows(bufE, d, n, np, s));
  TRY(zero_rows(bufH, 4 * d, n, np, s));
  // y = LN2(x1 + drop(f))
  TRY(launch_ln_bwd(dy, d, t.x1, d, t.f, d, lp.norm2_w, 1e-5f, n, d, bufB, d, G(&stlt_layer_params::norm2_w),
                    G(&stlt_layer_params::norm2_b), RED(sc), s, dr, site0 + 3, bufD, 0, G(&stlt_layer_params::lin2_b), rows));
  TRY(ffn_hidden_backward(df, lp.lin2_w, t.u, bufH, n, d, G(&stlt_layer_params::lin1_b), sc, dr, site0 + 2, rows, s));  // bufH = du
  TRY(dx_product(bufH, 4 * d, lp.lin1_w, 4 * d, d, bufB, d, bufC, d, n, sc, s));  // bufC = dx1 = du·W1 + ds2
  // x1 = LN1(x[rows] + drop(a)), a = ctx[rows]·Woᵀ + bo: gather the two inputs again (bufQ is free until the attention backward)
  float* g_x = bufQ;
  float* g_ctx = bufQ + np * d;
  TRY(launch_gather_rows(t.x, d, rows, n, d, g_x, s));
  TRY(launch_gather_rows(t.ctx, d, rows, n, d, g_ctx, s));
  TRY(launch_ln_bwd(bufC, d, g_x, d, t.a, d, lp.norm1_w, 1e-5f, n, d, ds1, d, G(&stlt_layer_params::norm1_w),
                    G(&stlt_layer_params::norm1_b), RED(sc), s, dr, site0 + 1, bufE, 0, G(&stlt_layer_params::out_proj_b), rows));  // ds1, da
  TRY(dx_product(da, d, lp.out_proj_w, d, d, nullptr, 0, bufC, d, n, sc, s));  // bufC = dctx of the picked rows
  // the weight gradients of the three Linears that ran on the picked rows: one grouped launch, before bufH / bufQ are reused
  const StltWeightGradItem items[3] = {{df, d, t.h, 4 * d, np, G(&stlt_layer_params::lin2_w)},
                                       {bufH, 4 * d, t.x1, d, np, G(&stlt_layer_params::lin1_w)},
                                       {da, d, g_ctx, d, np, G(&stlt_layer_params::out_proj_w)}};
  TRY(weight_grad_all(items, 3, sc, s));
  // the other rows' attention outputs were never read: their dctx is zero
  TRY(launch_scatter_rows(bufC, rows, n, d, bufH, M, s));                                           // bufH (as M x d) = dctx
  TRY(launch_attn_bwd(t.qkv, bufH, kpm, causal, S, L, H, d / H, bufQ, s, dr, site0, G(&stlt_layer_params::in_proj_b), RED(sc), rg));  // bufQ = dqkv
  TRY(weight_grad(bufQ, 3 * d, t.x, d, Mp, G(&stlt_layer_params::in_proj_w), sc, s));
  TRY(launch_scatter_rows(ds1, rows, n, d, bufC, M, s));                                            // residual path: ds1 on the picked rows only
  TRY(dx_product(bufQ, 3 * d, lp.in_proj_w, 3 * d, d, bufC, d, bufA, d, M, sc, s));  // bufA = dx = dqkv·Win + ds1
  // the 4d-wide view of bufH lost its zero rows past M to the dctx image only below M*d floats: nothing to restore
  return 0;
}

static int check_train(const stlt_params* p, const stlt_inputs* in, bool need_head = true) {
  if (!p || !in) return stlt_set_error(STLT_EINVAL, "null params/inputs");
  if (!stlt_heads_ok(p->d, p->H))
    return stlt_set_error(STLT_EINVAL, "hidden_size %lld / heads %lld: need hidden_size %% heads == 0, a head dim of at most 256 and hidden_size %% 4 == 0", (long long)p->d, (long long)p->H);
  if (p->n_spatial < 0 || p->n_spatial > 64 || p->n_temporal < 0 || p->n_temporal > 64) return stlt_set_error(STLT_EINVAL, "layer count out of range");
  if (in->B <= 0 || in->T <= 0 || in->N <= 0 || in->T > p->n_positions) return stlt_set_error(STLT_EINVAL, "bad batch shape");
  if (!in->categories || !in->boxes || !in->kpm_boxes || !in->frame_types || !in->kpm_frames || !in->lengths)
    return stlt_set_error(STLT_EINVAL, "null input tensor");
  if (need_head && (!p->fc1_w || !p->fc2_w || p->n_classes <= 0)) return stlt_set_error(STLT_EINVAL, "prediction head missing");
  if (p->n_categories > STLT_TRAIN_MAX_CATEGORIES)  // the embedding-gradient kernel keeps per-category sums in LDS (backward.hip: embed_bwd_kernel)
    return stlt_set_error(STLT_EINVAL, "training supports at most %d object categories (got %lld)", STLT_TRAIN_MAX_CATEGORIES, (long long)p->n_categories);
  return 0;
}

}  // namespace

extern "C" {

int stlt_set_train_side_stream(int on) { g_dw_side_wanted = on < 0 ? -1 : (on != 0); return 0; }  // < 0: back to STLT_TRAIN_DW_STREAM / the default
int stlt_get_train_side_stream(void) { return dw_side_wanted() ? 1 : 0; }

size_t stlt_train_tape_bytes(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_spatial, int64_t n_temporal) {
  if (B <= 0 || T <= 0 || N <= 0 || d <= 0 || n_spatial < 0 || n_spatial > 64 || n_temporal < 0 || n_temporal > 64) return 0;
  return tape_layout(nullptr, B, T, N, d, n_spatial, n_temporal).bytes;
}

size_t stlt_train_scratch_bytes(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_categories) {
  if (B <= 0 || T <= 0 || N <= 0 || d <= 0 || n_categories <= 0 || n_categories > STLT_TRAIN_MAX_CATEGORIES) return 0;
  return scratch_layout(nullptr, B, T, N, d, n_categories).bytes;
}

// Row counts of the ragged index: the caller's (stlt_inputs.n_real_tokens / n_real_frames, no synchronisation; `pad`: the forward makes the
// index safe for them, the backward finds it so), else one device->host copy + stream synchronisation.
static int read_ragged_counts(const RaggedIndex& ix, const stlt_inputs* in, bool pad, int64_t& Ms, int64_t& Mf, bool* from_host, hipStream_t s) {
  *from_host = in->n_real_tokens > 0 || in->n_real_frames > 0;
  if (*from_host) {
    Ms = in->n_real_tokens;
    Mf = in->n_real_frames;
    if (pad) return launch_ragged_host_counts(ix, Ms, Mf, in->B * in->T * in->N, in->B * in->T, s);
    if (Ms <= 0 || Mf <= 0 || Ms > in->B * in->T * in->N || Mf > in->B * in->T || Mf > Ms) return stlt_set_error(STLT_EINVAL, "skip-padding: n_real_tokens / n_real_frames do not fit the batch");
    return 0;
  }
  int counts[4] = {0, 0, 0, 0};
  if (hipError_t e = hipMemcpyAsync(counts, ix.counts, sizeof(counts), hipMemcpyDeviceToHost, s); e != hipSuccess)
    return stlt_set_error((int)e, "skip-padding: count read-back: %s", hipGetErrorString(e));
  if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess)
    return stlt_set_error((int)e, "skip-padding: count read-back: %s", hipGetErrorString(e));
  if (counts[2] != 0)
    return stlt_set_error(STLT_EINVAL, "skip-padding needs collater-shaped masks: slot 0 of every real frame unmasked and frame lengths-1 real (datasets.py:247-288)");
  Ms = counts[0];
  Mf = counts[1];
  return 0;
}

int stlt_train_forward(const stlt_params* p, const stlt_inputs* in, void* tape_mem, size_t tape_bytes, float* logits,
                       float dropout_p, uint64_t dropout_seed, int flags, stlt_stream_t stream) {
  const bool backbone_only = (flags & STLT_FLAG_TRAIN_BACKBONE) != 0;  // `logits` is then the (B*T, d) backbone output
  TRY(check_train(p, in, !backbone_only));
  if (backbone_only && (flags & STLT_FLAG_SKIP_PADDING)) return stlt_set_error(STLT_EINVAL, "STLT_FLAG_TRAIN_BACKBONE excludes STLT_FLAG_SKIP_PADDING");
  if (!logits || !tape_mem) return stlt_set_error(STLT_EINVAL, "stlt_train_forward: null logits/tape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = in->B, T = in->T, N = in->N, d = p->d, H = p->H;
  const Tape t = tape_layout((char*)tape_mem, B, T, N, d, p->n_spatial, p->n_temporal);
  if (tape_bytes < t.bytes) return stlt_set_error(STLT_EWORKSPACE, "tape %zu B < required %zu B", tape_bytes, t.bytes);
  StltGemmScratch gemm_scratch(t.sk, STLT_GEMM_SCRATCH_BYTES);
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return stlt_set_error(STLT_EINVAL, "dropout probability must be in [0,1)");
  const StltDrop dr = stlt_drop_make(dropout_p, dropout_seed);
  // Padded schedule: every (clip, frame, slot) row, masks applied inside the attention.  Skip-padding: the same
  // launches over the real rows only (ragged.hip); dropout masks are then drawn per compacted row.
  const bool ragged = (flags & STLT_FLAG_SKIP_PADDING) != 0;
  int64_t tok = B * T * N, BT = B * T;
  bool host_counts = false;
  const RaggedIndex ix = ragged_index_carve(t.ridx, B, T, N);
  if (ragged) {
    if (N > AB_MAX_ROWS || T > AB_MAX_ROWS)  // the reverse sweep would refuse the tape anyway; head dims other than 64 hold a segment's keys in LDS
      return stlt_set_error(STLT_EINVAL, "skip-padding training supports sequences of at most %d tokens (N=%lld, T=%lld)", AB_MAX_ROWS, (long long)N, (long long)T);
    TRY(launch_ragged_index(in->kpm_boxes, in->kpm_frames, in->lengths, B, T, N, ix, s));
    TRY(read_ragged_counts(ix, in, true, tok, BT, &host_counts, s));
  } else {
    TRY(launch_padded_rows(in->lengths, B, T, N, ix, s));  // rows the tail layers pick: f*N and b*T + lengths-1
  }
  // the last layer of each tower only has to produce the rows that are read afterwards (layer_forward_tail)
  const bool sp_tail = p->n_spatial > 0 && 2 * BT <= tok, tp_tail = !backbone_only && p->n_temporal > 0 && 2 * B <= BT;
  float* x0 = p->n_spatial > 0 ? t.sp[0].x : t.sp_out;
  TRY(launch_embed(in->categories, in->boxes, in->scores, p->cat_emb, p->n_categories, p->box_w, p->box_b, p->score_w,
                   p->score_b, p->emb_ln_w, p->emb_ln_b, p->ln_eps, tok, d, x0, s, t.s_embed, dr, ragged ? ix.t_orig : nullptr));
  for (int64_t l = 0; l < p->n_spatial; ++l) {
    const uint32_t site = (uint32_t)(8 * (l + 1));
    if (l == p->n_spatial - 1 && sp_tail) {
      TRY(layer_forward_tail(p->spatial[l], d, H, t.sp[l], tok, B * T, N, in->kpm_boxes, 0, STLT_K_ATTN_SPATIAL,
                             ragged ? ix.t_seg_start : nullptr, ragged ? ix.t_seg_end : nullptr, ix.f_cls_row, BT, t.sp_out, dr, site, s));
    } else {
      float* y = l + 1 < p->n_spatial ? t.sp[l + 1].x : t.sp_out;
      TRY(layer_forward(p->spatial[l], d, H, t.sp[l], tok, B * T, N, in->kpm_boxes, 0, STLT_K_ATTN_SPATIAL, y, dr, site, s,
                        ragged ? ix.t_seg_start : nullptr, ragged ? ix.t_seg_end : nullptr));
    }
  }
  float* tp_dst = backbone_only ? logits : t.tp_out;  // where the temporal tower's output rows land
  float* g0 = p->n_temporal > 0 ? t.tp[0].x : tp_dst;
  const float* cls = t.sp_out;  // (frames, d) when the tail layer ran, else the CLS rows inside the token buffer
  int64_t cls_stride = d;
  if (!sp_tail) {
    if (ragged) {  // CLS rows are not evenly strided: gather them (tp_out is free until the last temporal layer writes it)
      TRY(launch_gather_rows(t.sp_out, d, ix.f_cls_row, BT, d, t.tp_out, s));
      cls = t.tp_out;
    } else {
      cls_stride = N * d;
    }
  }
  TRY(launch_frames_embed(cls, cls_stride, in->frame_types, p->pos_emb, p->type_emb, p->frames_ln_w, p->frames_ln_b, p->ln_eps, B, T,
                          d, g0, s, t.s_frames, dr, ragged ? ix.f_orig : nullptr, BT));
  for (int64_t l = 0; l < p->n_temporal; ++l) {
    const uint32_t site = (uint32_t)(8 * (p->n_spatial + l + 1));
    if (l == p->n_temporal - 1 && tp_tail) {
      TRY(layer_forward_tail(p->temporal[l], d, H, t.tp[l], BT, B, T, in->kpm_frames, 1, STLT_K_ATTN_TEMPORAL,
                             ragged ? ix.f_seg_start : nullptr, ragged ? ix.f_seg_end : nullptr, ix.last_row, B, t.h0, dr, site, s));
    } else {
      float* y = l + 1 < p->n_temporal ? t.tp[l + 1].x : tp_dst;
      TRY(layer_forward(p->temporal[l], d, H, t.tp[l], BT, B, T, in->kpm_frames, 1, STLT_K_ATTN_TEMPORAL, y, dr, site, s,
                        ragged ? ix.f_seg_start : nullptr, ragged ? ix.f_seg_end : nullptr));
    }
  }
  if (backbone_only) return 0;
  if (!tp_tail) TRY(launch_gather_rows(t.tp_out, d, ix.last_row, B, d, t.h0, s));                    // models.py:189-192
  TRY(launch_linear(t.h0, d, p->fc1_w, p->fc1_b, t.u0, d, B, d, d, STLT_ACT_NONE, s));
  TRY(launch_gelu_fwd(t.u0, t.z1, B * d, s));
  TRY(launch_add_layernorm(t.z1, d, nullptr, 0, p->head_ln_w, p->head_ln_b, p->ln_eps, B, d, t.z2, d, s));
  TRY(launch_linear(t.z2, d, p->fc2_w, p->fc2_b, logits, p->n_classes, B, p->n_classes, d, STLT_ACT_NONE, s));
  // the caller's row counts were taken on trust: NaN logits (hence a NaN loss) when they are not the index's or the masks break the contract
  return host_counts ? launch_ragged_poison(ix, tok, BT, false, logits, B * p->n_classes, s) : 0;
}

int stlt_train_backward(const stlt_params* p, const stlt_params* g, const stlt_inputs* in, const void* tape_mem,
                        size_t tape_bytes, void* scratch_mem, size_t scratch_bytes, const float* dlogits,
                        float dropout_p, uint64_t dropout_seed, int flags, stlt_ctx* ctx, stlt_stream_t stream) {
  const bool backbone_only = (flags & STLT_FLAG_TRAIN_BACKBONE) != 0;  // `dlogits` is then the gradient of the (B*T, d) backbone output
  TRY(check_train(p, in, !backbone_only));
  if (backbone_only && (flags & STLT_FLAG_SKIP_PADDING)) return stlt_set_error(STLT_EINVAL, "STLT_FLAG_TRAIN_BACKBONE excludes STLT_FLAG_SKIP_PADDING");
  if (!g || !dlogits || !tape_mem || !scratch_mem) return stlt_set_error(STLT_EINVAL, "stlt_train_backward: null argument");
  hipStream_t s = (hipStream_t)stream;
  StltCtxScope ctx_scope(ctx, s);  // the sweep's input-gradient products may read the context's transposed weight copies
  if (ctx_scope.error()) return ctx_scope.error();
  const int64_t B = in->B, T = in->T, N = in->N, d = p->d, H = p->H, K = p->n_classes;
  const Tape t = tape_layout((char*)const_cast<void*>(tape_mem), B, T, N, d, p->n_spatial, p->n_temporal);
  if (tape_bytes < t.bytes) return stlt_set_error(STLT_EWORKSPACE, "tape %zu B < required %zu B", tape_bytes, t.bytes);
  Scratch sc = scratch_layout((char*)scratch_mem, B, T, N, d, p->n_categories);
  if (scratch_bytes < sc.bytes) return stlt_set_error(STLT_EWORKSPACE, "scratch %zu B < required %zu B", scratch_bytes, sc.bytes);
  // the sweep's ~55 partial-row reductions (LayerNorm / bias gradients) are collected and run as a few batched launches
  // (STLT_TRAIN_DEFER_REDUCE=0: one launch each, A/B runs)
  const bool defer_on = defer_wanted();
  StltReduceDefer defer;
  defer.s = (hipStream_t)stream;
  defer.pool = sc.red_pool;
  defer.pool_floats = sc.red_pool_floats;
  struct DeferGuard {
    StltReduceDefer* d;
    explicit DeferGuard(StltReduceDefer* x) : d(x) { stlt_reduce_defer_set(d); }
    ~DeferGuard() { if (d && d->n > 0) (void)stlt_reduce_defer_flush(d); stlt_reduce_defer_set(nullptr); }  // error exits: what the producers left is still reduced
  } defer_guard(defer_on && sc.red_pool ? &defer : nullptr);
  if (defer_on && sc.red_pool) sc.defer = &defer;
  int64_t tok = B * T * N, BT = B * T;
  int64_t tokp = t.tokp, btp = t.btp;
  StltGemmScratch gemm_scratch(sc.sk, STLT_GEMM_SCRATCH_BYTES);
  const StltDrop dr = stlt_drop_make(dropout_p, dropout_seed);
  auto W = [](const float* q) { return const_cast<float*>(q); };
  // Skip-padding: the tape holds the real rows only (same buffers, fewer rows).  The weight-gradient products
  // contract over the row count rounded up to 32, so the gradient-side operands' rows between the count and its
  // round-up are zeroed here (the padded schedule never dirties them; a ragged row count changes every step).
  const bool ragged = (flags & STLT_FLAG_SKIP_PADDING) != 0;
  const bool do_upper = !(flags & STLT_FLAG_TRAIN_LOWER_ONLY), do_lower = !(flags & STLT_FLAG_TRAIN_UPPER_ONLY);
  if (!do_upper && !do_lower) return stlt_set_error(STLT_EINVAL, "stlt_train_backward: UPPER_ONLY and LOWER_ONLY exclude each other");
  const RaggedIndex ix = ragged_index_carve(t.ridx, B, T, N);  // filled by the forward (ragged index, or the padded layout's picked rows)
  AttnBwdRagged rg_sp{}, rg_tp{};
  if (ragged) {
    bool host_counts = false;
    TRY(read_ragged_counts(ix, in, false, tok, BT, &host_counts, s));
    tokp = up32(tok);
    btp = up32(BT);
    if (N > AB_MAX_ROWS || T > AB_MAX_ROWS) return stlt_set_error(STLT_EINVAL, "attention backward supports sequences of at most %d tokens", AB_MAX_ROWS);
    const int fpg = N <= 32 ? (int)(32 / N) : 1;  // whole frames per attention-backward group
    TRY(launch_ragged_groups(ix, tok, BT, fpg, s));
    rg_sp = AttnBwdRagged{ix.sp_grp_ptr, ix.t_seg_start, ix.t_seg_end, (BT + fpg - 1) / fpg, tok, (int)(fpg * N)};
    rg_tp = AttnBwdRagged{ix.clip_frm_off, ix.f_seg_start, ix.f_seg_end, B, BT, (int)T};
  }
  // Both schedules: the caller may hand in scratch that an earlier step with another (B,T,N) left dirty (two shapes
  // can round to the same byte count), so the rows the weight-gradient products read beyond the row count are
  // cleared every step: at most 31 rows per buffer.
  // (the fusion models' layout branch — STLT_FLAG_TRAIN_BACKBONE — keeps one stream: measured 43.75 against 44.1 ms per CACNF step at 64
  // clips; the block-level calls around the sweep are single-stream and the side launches only delay their small kernels)
  DwSide side = backbone_only ? DwSide{} : dw_side_open(sc, ctx);
  DwSideHold side_hold(&side, s);
  // weight-gradient queues: the spatial tower flushes per layer over two sets; the temporal tower, when it has few rows, collects up to
  // eight layers (32 products) per grouped launch over eight sets (STLT_TRAIN_DW_GROUP_LAYERS=1: per layer, A/B runs)
  static const int group_env = [] { const char* e = getenv("STLT_TRAIN_DW_GROUP_LAYERS"); return e ? atoi(e) : 8; }();
  DwQueue q_sp, q_tp;
  q_sp.side = q_tp.side = &side;
  q_sp.sets[0] = GradBufs{sc.sB, sc.sD, sc.sE, sc.sQKV, sc.sH}; q_sp.sets[1] = sc.s2; q_sp.n_sets = sc.s2.B ? 2 : 1;  // one stream: a layer's products are flushed before the next layer rewrites the set
  q_tp.sets[0] = GradBufs{sc.tB, sc.tD, sc.tE, sc.tQKV, sc.tH}; q_tp.sets[1] = sc.t2; q_tp.n_sets = 2;
  if (sc.n_tx > 0 && group_env > 1 && sc.sk) {
    for (int i = 0; i < sc.n_tx; ++i) q_tp.sets[2 + i] = sc.tx[i];
    q_tp.n_sets = 2 + sc.n_tx;
    q_tp.group_layers = group_env < q_tp.n_sets ? group_env : q_tp.n_sets;
  }
  if (do_lower) {
    for (int k = 0; k < q_sp.n_sets; ++k) {
      for (float* b : {q_sp.sets[k].B, q_sp.sets[k].D, q_sp.sets[k].E}) TRY(zero_rows(b, d, tok, tokp, s));
      TRY(zero_rows(q_sp.sets[k].Q, 3 * d, tok, tokp, s));
      TRY(zero_rows(q_sp.sets[k].H, 4 * d, tok, tokp, s));
    }
  }
  if (do_upper) {
    for (int k = 0; k < q_tp.n_sets; ++k) {
      for (float* b : {q_tp.sets[k].B, q_tp.sets[k].D, q_tp.sets[k].E}) TRY(zero_rows(b, d, BT, btp, s));
      TRY(zero_rows(q_tp.sets[k].Q, 3 * d, BT, btp, s));
      TRY(zero_rows(q_tp.sets[k].H, 4 * d, BT, btp, s));
    }
  }
  const bool sp_tail = p->n_spatial > 0 && 2 * BT <= tok, tp_tail = !backbone_only && p->n_temporal > 0 && 2 * B <= BT;  // as the forward decided
  if (do_upper && backbone_only) {
    // the gradient of every row of the temporal tower's output arrives from the fusion layers: no head, no row gather
    if (hipError_t e = hipMemcpyAsync(sc.tA, dlogits, (size_t)BT * d * sizeof(float), hipMemcpyDeviceToDevice, s); e != hipSuccess)
      return stlt_set_error((int)e, "train_backward: copy: %s", hipGetErrorString(e));
    TRY(zero_rows(sc.tA, d, BT, btp, s));
    for (int64_t l = p->n_temporal - 1; l >= 0; --l)
      TRY(layer_backward(p->temporal[l], g->temporal ? &g->temporal[l] : nullptr, t.tp[l], d, H, BT, btp, B, T, in->kpm_frames, 1,
                         sc.tA, sc.tB, sc.tC, sc.tD, sc.tE, sc.tQKV, sc.tH, sc, dr, (uint32_t)(8 * (p->n_spatial + l + 1)), s, nullptr, &q_tp));
    TRY(dwq_flush(q_tp, sc, s));
  } else if (do_upper) {
  // ---- prediction head (models.py:162-163): logits = z2·W2ᵀ+b2, z2 = LN(z1), z1 = gelu(u0), u0 = h0·W1ᵀ+b1
  if (g->fc2_w) TRY(launch_small_gemm(dlogits, 1, K, t.z2, d, 1, W(g->fc2_w), d, K, d, B, 1, s));   // (K,d) += dlogitsᵀ·z2
  if (g->fc2_b) TRY(launch_colsum_acc(dlogits, K, B, K, W(g->fc2_b), RED(sc), s));
  TRY(launch_small_gemm(dlogits, K, 1, p->fc2_w, d, 1, sc.hA, d, B, d, K, 0, s));                   // hA = dz2
  TRY(launch_ln_bwd(sc.hA, d, t.z1, d, nullptr, 0, p->head_ln_w, p->ln_eps, B, d, sc.hB, d, W(g->head_ln_w), W(g->head_ln_b),
                    RED(sc), s));                                                                    // hB = dz1
  TRY(launch_gelu_bwd(sc.hB, t.u0, sc.hB, B * d, s));                                               // hB = du0
  // the two d x d products of fc1 go to the MFMA kernel (it contracts over multiples of 32 clips; the strided kernel takes the rest)
  if (g->fc1_w) {                                                                                   // (d,d) += du0ᵀ·h0
    const int64_t Bf = B / 32 * 32;
    if (Bf > 0) TRY(launch_gemm(1, 1, sc.hB, d, t.h0, d, nullptr, W(g->fc1_w), d, W(g->fc1_w), d, 0, d, d, Bf, 1, STLT_ACT_NONE, s));
    if (B > Bf) TRY(launch_small_gemm(sc.hB + Bf * d, 1, d, t.h0 + Bf * d, d, 1, W(g->fc1_w), d, d, d, B - Bf, 1, s));
  }
  if (g->fc1_b) TRY(launch_colsum_acc(sc.hB, d, B, d, W(g->fc1_b), RED(sc), s));
  TRY(launch_gemm(0, 1, sc.hB, d, p->fc1_w, d, nullptr, nullptr, 0, sc.hA, d, 0, B, d, d, 1, STLT_ACT_NONE, s));  // hA = dh0
  // ---- temporal transformer
  int64_t l_tp = p->n_temporal - 1;
  if (tp_tail) {
    TRY(layer_backward_tail(p->temporal[l_tp], g->temporal ? &g->temporal[l_tp] : nullptr, t.tp[l_tp], d, H, BT, btp, B, T, in->kpm_frames, 1,
                            ix.last_row, B, sc.hA, sc.tA, sc.tB, sc.tC, sc.tD, sc.tE, sc.tQKV, sc.tH, sc, dr,
                            (uint32_t)(8 * (p->n_spatial + l_tp + 1)), s, ragged ? &rg_tp : nullptr));
    --l_tp;
  } else {
    TRY(launch_scatter_rows(sc.hA, ix.last_row, B, d, sc.tA, btp, s));                              // tA = d(backbone out)
  }
  for (int64_t l = l_tp; l >= 0; --l)
    TRY(layer_backward(p->temporal[l], g->temporal ? &g->temporal[l] : nullptr, t.tp[l], d, H, BT, btp, B, T, in->kpm_frames, 1,
                       sc.tA, sc.tB, sc.tC, sc.tD, sc.tE, sc.tQKV, sc.tH, sc, dr, (uint32_t)(8 * (p->n_spatial + l + 1)), s,
                       ragged ? &rg_tp : nullptr, &q_tp));
    TRY(dwq_flush(q_tp, sc, s));
  }  // upper half: sc.tA now holds the gradient wrt the temporal tower's input
  if (!do_lower) { TRY(stlt_reduce_defer_flush(sc.defer)); return dw_side_join(&side, s); }
  // ---- frames embeddings (models.py:98-111).  The gradient wrt the frames' CLS rows goes to tC, a chain-only buffer: the temporal
  // tower's last weight-gradient products may still be reading tB / tD / tE / tQKV / tH on the side stream.
  float* d_cls = sc.tC;
  TRY(launch_ln_bwd(sc.tA, d, t.s_frames, d, nullptr, 0, p->frames_ln_w, p->ln_eps, BT, d, d_cls, d, W(g->frames_ln_w),
                    W(g->frames_ln_b), RED(sc), s, dr, 0, nullptr, STLT_SITE_FRAMES));                // d_cls = gradient wrt the frames' CLS rows
  const bool dense_scatter = !ragged && !sp_tail;  // padded dense schedule: the CLS rows sit at stride N in the token buffer
  TRY(launch_frames_bwd(d_cls, in->frame_types, B, T, N, d, dense_scatter ? sc.sA : nullptr, W(g->pos_emb), W(g->type_emb), RED(sc), s,
                        ragged ? ix.f_row_of : nullptr));
  // ---- spatial transformer
  int64_t l_sp = p->n_spatial - 1;
  if (sp_tail) {
    TRY(layer_backward_tail(p->spatial[l_sp], g->spatial ? &g->spatial[l_sp] : nullptr, t.sp[l_sp], d, H, tok, tokp, B * T, N, in->kpm_boxes, 0,
                            ix.f_cls_row, BT, d_cls, sc.sA, sc.sB, sc.sC, sc.sD, sc.sE, sc.sQKV, sc.sH, sc, dr, (uint32_t)(8 * (l_sp + 1)), s,
                            ragged ? &rg_sp : nullptr));
    --l_sp;
  } else if (ragged) {
    TRY(launch_scatter_rows(d_cls, ix.f_cls_row, BT, d, sc.sA, tokp, s));                            // sA = d(spatial out), CLS rows only
  }
  for (int64_t l = l_sp; l >= 0; --l)
    TRY(layer_backward(p->spatial[l], g->spatial ? &g->spatial[l] : nullptr, t.sp[l], d, H, tok, tokp, B * T, N, in->kpm_boxes, 0,
                       sc.sA, sc.sB, sc.sC, sc.sD, sc.sE, sc.sQKV, sc.sH, sc, dr, (uint32_t)(8 * (l + 1)), s, ragged ? &rg_sp : nullptr, &q_sp));
  // ---- category / box / score embeddings (models.py:29-39); sC for the same reason as tC above
  TRY(launch_ln_bwd(sc.sA, d, t.s_embed, d, nullptr, 0, p->emb_ln_w, p->ln_eps, tok, d, sc.sC, d, W(g->emb_ln_w), W(g->emb_ln_b),
                    RED(sc), s, dr, 0, nullptr, STLT_SITE_EMBED));
  TRY(launch_embed_bwd(sc.sC, in->categories, in->boxes, in->scores, p->n_categories, tok, d, W(g->cat_emb), W(g->box_w),
                       W(g->box_b), W(g->score_w), W(g->score_b), RED(sc), s, ragged ? ix.t_orig : nullptr));
  TRY(stlt_reduce_defer_flush(sc.defer));
  return dw_side_join(&side, s);
}

}  // extern "C"
