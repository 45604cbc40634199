// Block-level training entry points: one native forward and one native reverse call per residual block of the fusion
// models (reference src/modelling/models.py:328-431 — SelfAttentionLayer / CrossAttentionLayer / the feed-forward block —
// and the two halves of nn.TransformerEncoderLayer as configured at :46-52, :118-124, :239-246), so that a CAF / CACNF /
// LCF optimisation step is a few dozen calls through the C-ABI instead of a few hundred op-level ones.  Pure orchestration
// of launchers the STLT path already has; the caller owns the tape tensors and the scratch.
//   attention block:  out = LN_eps( x + drop( MHA(x, c, c; kpm, causal, prob-dropout) · Woᵀ + bo ) )      c = x for self-attention
//   feed-forward block: out = LN_eps( x + drop( W2 · drop_inner( act(W1 x + b1) ) + b2 ) )                 act = GELU | ReLU
// Dropout sites: site0 = attention probabilities / inner dropout, site0 + 1 = the dropout in front of the residual.
#include "ctx.h"

namespace {

#define TRY(expr) do { int _e = (expr); if (_e) return _e; } while (0)
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// A block backward's device memory comes in two buffers.  `keep` holds what the block's QUEUED weight-gradient products read (their dY
// operands): while a context defers them (stlt_ctx_dw_defer) the caller keeps this buffer — a few rows x d floats — alive until the flush.
// `work` holds everything that is dead when the call returns its launches (stream-K partial tiles, the attention context gradient, reduction
// pools, a transposed weight for the split-bf16 products): one buffer per stream serves every block.
struct BlockScratch {
  // work
  char* lin;        // stlt_linear_bwd's scratch (stream-K partial tiles + column-sum partials)
  size_t lin_bytes;
  float *dctx, *red;
  float* red_pool;  // 4 x red: the call's partial-row reductions are deferred into one batched launch (StltReduceDefer)
  size_t red_floats;
  float* wt;        // 4 d^2 floats: a transposed weight for the opt-in split-bf16 input-gradient products
  size_t wt_floats, work_total;
  // keep
  float *ds, *da, *dq, *dkv, *dh;
  size_t keep_total;
};

enum { BLOCK_ATTN = 0, BLOCK_FFN = 1 };

// rows = the larger of the query-side and key-side row counts
BlockScratch block_scratch(char* keep, char* work, int64_t rows, int64_t d, int kind) {
  BlockScratch b;
  const size_t f = sizeof(float);
  size_t off = 0;
  char* base = work;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = align256(off + bytes); return p; };
  b.lin_bytes = stlt_linear_bwd_scratch_bytes(4 * d);
  b.lin = take(b.lin_bytes);
  b.dctx = (float*)take(kind == BLOCK_ATTN ? (size_t)rows * d * f : 0);
  int64_t red = ln_bwd_scratch_floats(d);
  if (512 * 4 * d > red) red = 512 * 4 * d;
  b.red = (float*)take((size_t)red * f);
  b.red_floats = (size_t)red;
  b.red_pool = (float*)take((size_t)red * 4 * f);
  b.wt_floats = (size_t)4 * d * d;
  b.wt = (float*)take(b.wt_floats * f);
  b.work_total = off;
  off = 0;
  base = keep;
  b.ds = (float*)take((size_t)rows * d * f);
  b.da = (float*)take((size_t)rows * d * f);
  b.dq = (float*)take(kind == BLOCK_ATTN ? (size_t)rows * 3 * d * f : 0);
  b.dkv = (float*)take(kind == BLOCK_ATTN ? (size_t)rows * 2 * d * f : 0);
  b.dh = (float*)take(kind == BLOCK_FFN ? (size_t)rows * 4 * d * f : 0);
  b.keep_total = off;
  return b;
}

// The partial-row reductions of one backward call (LayerNorm parameters, bias column sums) run as one batched launch at its end.
struct BlockDefer {
  StltReduceDefer d;
  BlockDefer(const BlockScratch& sc, hipStream_t s) {
    d.s = s; d.pool = sc.red_pool; d.pool_floats = sc.red_floats * 4;
    stlt_reduce_defer_set(sc.red_pool ? &d : nullptr);
  }
  ~BlockDefer() { stlt_reduce_defer_set(nullptr); }
  float* chunk(const BlockScratch& sc) { int err = 0; return stlt_reduce_defer_chunk(sc.red_pool ? &d : nullptr, sc.red_floats, sc.red, &err); }
  int flush() { return stlt_reduce_defer_flush(&d); }
};

// The weight-gradient products of a block are collected and run as ONE grouped stream-K launch at the end of the block's
// backward (gemm.hip: launch_weight_grad_group) when every product contracts over a multiple of 32 rows; otherwise product
// by product (MFMA over the 32-row multiple + the strided kernel for the rest: stlt_linear_bwd).
struct DwList {
  StltWeightGradItem it[4];
  int n = 0;
  void add(const float* dy, int64_t n_out, const float* x, int64_t k_in, int64_t rows, float* g_w) {
    if (g_w && rows > 0) it[n++] = StltWeightGradItem{dy, n_out, x, k_in, rows, g_w};
  }
};
// Collector (stlt_ctx_dw_defer): while the call's context has it on, a block's grouped weight-gradient launch is not made at the end of the
// block's backward — its products are queued in the context and stlt_ctx_dw_flush runs the queue as grouped launches of up to 32 products.  A
// block's own group is 2 - 4 products over 2048 / 2112 rows (18 k-steps per workgroup: 0.39 of the MFMA peak, 34 launches + 34 fix-ups per
// CACNF step); 32 products per launch run at the rate of the STLT sweep's 8-layer groups (0.84).  The caller keeps every operand alive until the
// flush.  The queue lives in the CONTEXT the block call names, not in a thread or the process: torch's autograd engine runs the block backwards
// on its own device thread while the trainer switches the collector and flushes it from the thread that called backward() (a thread-local
// collector queued the products on the engine's thread, where nobody ever flushed them: caught by the GPU suite as all-zero weight gradients in
// a later test); two training loops in one process hold two contexts and never see each other's products.
int flush_dw(const DwList& l, const BlockScratch& sc, hipStream_t s) {
  if (l.n == 0) return 0;
  bool group_ok = true;
  for (int i = 0; i < l.n; ++i) group_ok = group_ok && l.it[i].rows % 32 == 0 && l.it[i].rows <= 4096;  // long contractions: separate launches are as fast (train.hip: weight_grad_all)
  if (stlt_ctx* c = group_ok ? stlt_ctx_current() : nullptr) {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->dw_on && c->dw_n + l.n <= STLT_DW_DEFER_CAP) {
      for (int i = 0; i < l.n; ++i) c->dw[c->dw_n++] = l.it[i];
      return 0;
    }
  }
  StltGemmScratch lend(sc.lin, STLT_GEMM_SCRATCH_BYTES);
  if (group_ok) return launch_weight_grad_group(l.it, l.n, s);
  for (int i = 0; i < l.n; ++i)
    TRY(stlt_linear_bwd(l.it[i].x, l.it[i].g_w /* unused: dx is null */, l.it[i].dy, l.it[i].rows, l.it[i].n_out, l.it[i].k_in, nullptr, l.it[i].g_w, nullptr,
                        nullptr, sc.lin, sc.lin_bytes, (stlt_stream_t)s));
  return 0;
}

// dx (M,K) = dy·W (+ add); the weight gradient dw (N,K) += dyᵀ·x is queued on `dws`; db (N) += column sums of dy.
// N (the contraction length of dx) is a multiple of 32 for every Linear of these blocks (d, 2d, 3d, 4d with d = 64 H).
int linear_bwd(const float* x, const float* w, const float* dy, int64_t M, int64_t N, int64_t K, float* dx, const float* add, float* dw,
               float* db, DwList& dws, const BlockScratch& sc, hipStream_t s, BlockDefer* defer = nullptr) {
  if (dx && M > 0) {
    StltGemmScratch lend(sc.lin, STLT_GEMM_SCRATCH_BYTES);
    bool taken = false;
    if ((size_t)(N * K) <= sc.wt_floats) TRY(launch_input_grad_bf16x3(dy, N, w, N, K, add, K, dx, K, M, sc.wt, s, &taken));
    if (!taken) TRY(launch_input_grad_gemm16(dy, N, w, N, K, add, K, dx, K, M, s, &taken));  // few rows: whole small tiles
    if (!taken) TRY(launch_gemm(0, 1, dy, N, w, K, nullptr, add, K, dx, K, 0, M, K, N, 1, STLT_ACT_NONE, s));
  }
  dws.add(dy, N, x, K, M, dw);
  if (db && M > 0) return launch_colsum_acc(dy, N, M, N, db, sc.red, s);
  return 0;
}

}  // namespace

extern "C" {

int stlt_ctx_dw_defer(stlt_ctx* c, int mode) {
  if (!stlt_ctx_valid(c)) return stlt_set_error(STLT_EINVAL, "stlt_ctx_dw_defer: not a live context handle");
  std::lock_guard<std::mutex> lk(c->mu);
  if (mode == 1) { c->dw_on = true; return 0; }
  if (mode == 0) { c->dw_on = false; return 0; }
  if (mode == -1) { c->dw_on = false; c->dw_n = 0; return 0; }
  return stlt_set_error(STLT_EINVAL, "stlt_ctx_dw_defer: mode 1 (collect), 0 (stop collecting; queued products stay) or -1 (stop and discard)");
}
int stlt_ctx_dw_pending(stlt_ctx* c) {
  if (!stlt_ctx_valid(c)) return -1;
  std::lock_guard<std::mutex> lk(c->mu);
  return c->dw_n;
}
int stlt_ctx_dw_flush(stlt_ctx* c, void* gemm_scratch, size_t gemm_scratch_bytes, stlt_stream_t stream) {
  if (!stlt_ctx_valid(c)) return stlt_set_error(STLT_EINVAL, "stlt_ctx_dw_flush: not a live context handle");
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->dw_n == 0) return 0;
  if (!gemm_scratch || gemm_scratch_bytes < STLT_GEMM_SCRATCH_BYTES) return stlt_set_error(STLT_EWORKSPACE, "stlt_ctx_dw_flush: stlt_gemm_scratch_bytes() of scratch are required");
  StltGemmScratch lend(gemm_scratch, STLT_GEMM_SCRATCH_BYTES);
  hipStream_t s = (hipStream_t)stream;
  int rc = 0, i0 = 0;
  while (i0 < c->dw_n && rc == 0) {
    int i1 = i0;
    for (; i1 < c->dw_n && i1 - i0 < STLT_GEMM_GROUP_MAX; ++i1) {
      const float* lo = c->dw[i1].g_w;
      const float* hi = lo + c->dw[i1].n_out * c->dw[i1].k_in;
      bool clash = false;  // two products into overlapping gradient ranges (a weight shared by two blocks) never share a launch
      for (int j = i0; j < i1 && !clash; ++j) {
        const float* lo2 = c->dw[j].g_w;
        const float* hi2 = lo2 + c->dw[j].n_out * c->dw[j].k_in;
        clash = lo < hi2 && lo2 < hi;
      }
      if (clash) break;
    }
    rc = launch_weight_grad_group(c->dw + i0, i1 - i0, s);
    i0 = i1;
  }
  c->dw_n = 0;
  return rc;
}
size_t stlt_block_keep_bytes(int64_t rows, int64_t d, int kind) {
  if (rows <= 0 || d <= 0 || (kind != BLOCK_ATTN && kind != BLOCK_FFN)) return 0;
  return block_scratch(nullptr, nullptr, rows, d, kind).keep_total;
}
size_t stlt_block_work_bytes(int64_t rows, int64_t d) {
  if (rows <= 0 || d <= 0) return 0;
  return block_scratch(nullptr, nullptr, rows, d, BLOCK_ATTN).work_total;  // the attention block's is the larger layout
}

int stlt_attn_block_fwd_train(const stlt_attn_block_params* p, int64_t d, int64_t H, float eps, const float* x, int64_t Lq, const float* c,
                              int64_t Lk, const uint8_t* kpm, int causal, int64_t S, float drop_p, uint64_t seed, uint32_t site0, float* q,
                              float* kv, float* ctx, float* a, float* out, void* gemm_scratch, size_t gemm_scratch_bytes, stlt_stream_t stream) {
  if (!p || !x || !kpm || !q || !ctx || !a || !out || (c && !kv)) return stlt_set_error(STLT_EINVAL, "stlt_attn_block_fwd_train: null argument");
  if (gemm_scratch && gemm_scratch_bytes < STLT_GEMM_SCRATCH_BYTES) return stlt_set_error(STLT_EWORKSPACE, "stlt_attn_block_fwd_train: gemm scratch too small");
  StltGemmScratch lend(gemm_scratch, STLT_GEMM_SCRATCH_BYTES);  // under-filled products run as stream-K when scratch is lent
  if (!stlt_heads_ok(d, H)) return stlt_set_error(STLT_EINVAL, "stlt_attn_block_fwd_train: need d %% H == 0, a head dim of at most 256 and d %% 4 == 0");
  if (!(drop_p >= 0.f && drop_p < 1.f)) return stlt_set_error(STLT_EINVAL, "dropout probability must be in [0,1)");
  if (!c && Lq != Lk) return stlt_set_error(STLT_EINVAL, "stlt_attn_block_fwd_train: self-attention needs Lq == Lk");
  hipStream_t s = (hipStream_t)stream;
  const StltDrop dr = stlt_drop_make(drop_p, seed);
  const int64_t Mq = S * Lq, Mk = S * Lk;
  const int kid = causal ? STLT_K_ATTN_TEMPORAL : STLT_K_ATTN_SPATIAL;
  if (!c) {  // packed q | k | v
    TRY(launch_linear(x, d, p->in_proj_w, p->in_proj_b, q, 3 * d, Mq, 3 * d, d, STLT_ACT_NONE, s));
    TRY(launch_attn(q, kpm, causal, S, Lq, H, d / H, ctx, kid, s, dr, site0));
  } else {   // q from x, k | v from the other modality's tokens (models.py:362-382)
    TRY(launch_linear(x, d, p->in_proj_w, p->in_proj_b, q, d, Mq, d, d, STLT_ACT_NONE, s));
    TRY(launch_linear(c, d, p->in_proj_w + d * d, p->in_proj_b + d, kv, 2 * d, Mk, 2 * d, d, STLT_ACT_NONE, s));
    TRY(launch_attn_general(q, d, kv, kv + d, 2 * d, kpm, causal, S, Lq, Lk, H, d / H, ctx, kid, s, dr, site0));
  }
  TRY(launch_linear(ctx, d, p->out_proj_w, p->out_proj_b, a, d, Mq, d, d, STLT_ACT_NONE, s));
  return launch_add_layernorm(a, d, x, d, p->ln_w, p->ln_b, eps, Mq, d, out, d, s, dr, site0 + 1);
}

// g: gradient buffers of the block's parameters (same struct; ACCUMULATED into; null members are skipped).
// dx (S*Lq, d) = gradient wrt x; dc (S*Lk, d) = gradient wrt the context tokens (cross-attention only, may be null).
int stlt_attn_block_bwd_train(const stlt_attn_block_params* p, const stlt_attn_block_params* g, int64_t d, int64_t H, float eps, const float* x,
                              int64_t Lq, const float* c, int64_t Lk, const uint8_t* kpm, int causal, int64_t S, float drop_p, uint64_t seed,
                              uint32_t site0, const float* q, const float* kv, const float* ctx, const float* a, const float* dy, float* dx,
                              float* dc, stlt_ctx* tctx, void* keep, size_t keep_bytes, void* work, size_t work_bytes, stlt_stream_t stream) {
  if (!p || !g || !x || !kpm || !q || !ctx || !a || !dy || !dx || !keep || !work || (c && !kv)) return stlt_set_error(STLT_EINVAL, "stlt_attn_block_bwd_train: null argument");
  if (!stlt_heads_ok(d, H)) return stlt_set_error(STLT_EINVAL, "stlt_attn_block_bwd_train: need d %% H == 0, a head dim of at most 256 and d %% 4 == 0");
  if (S < 0 || Lq <= 0 || Lk <= 0 || S > (int64_t)0x7fffffff / (Lq > Lk ? Lq : Lk)) return stlt_set_error(STLT_EINVAL, "stlt_attn_block_bwd_train: bad shape (S %lld, Lq %lld, Lk %lld)", (long long)S, (long long)Lq, (long long)Lk);
  if (S == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int64_t Mq = S * Lq, Mk = S * Lk;
  const BlockScratch sc = block_scratch((char*)keep, (char*)work, Mq > Mk ? Mq : Mk, d, BLOCK_ATTN);
  if (keep_bytes < sc.keep_total) return stlt_set_error(STLT_EWORKSPACE, "stlt_attn_block_bwd_train: keep buffer %zu B < required %zu B", keep_bytes, sc.keep_total);
  if (work_bytes < sc.work_total) return stlt_set_error(STLT_EWORKSPACE, "stlt_attn_block_bwd_train: work buffer %zu B < required %zu B", work_bytes, sc.work_total);
  StltCtxScope scope(tctx, s);
  if (scope.error()) return scope.error();
  const StltDrop dr = stlt_drop_make(drop_p, seed);
  auto G = [&](const float* q_) { return const_cast<float*>(q_); };
  BlockDefer defer(sc, s);
  // out = LN(x + drop(a)): ds = residual-path gradient, da = gradient wrt a (= ds without dropout); out_proj_b += colsum(da)
  float* da = dr.thr ? sc.da : sc.ds;
  TRY(launch_ln_bwd(dy, d, x, d, a, d, p->ln_w, eps, Mq, d, sc.ds, d, G(g->ln_w), G(g->ln_b), defer.chunk(sc), s, dr, site0 + 1, sc.da, 0, G(g->out_proj_b)));
  // a = ctx·Woᵀ + bo
  DwList dws;
  TRY(linear_bwd(ctx, p->out_proj_w, da, Mq, d, d, sc.dctx, nullptr, G(g->out_proj_w), nullptr, dws, sc, s));
  if (!c) {
    // packed self-attention: dqkv (Mq, 3d); in_proj_b += its column sums (accumulated by the attention backward)
    TRY(launch_attn_bwd(q, sc.dctx, kpm, causal, S, Lq, H, d / H, sc.dq, s, dr, site0, G(g->in_proj_b), defer.chunk(sc)));
    TRY(linear_bwd(x, p->in_proj_w, sc.dq, Mq, 3 * d, d, dx, sc.ds, G(g->in_proj_w), nullptr, dws, sc, s));  // dx = dqkv·Win + ds
    TRY(defer.flush());
    return flush_dw(dws, sc, s);
  }
  TRY(stlt_attn_bwd(q, d, kv, kv + d, 2 * d, sc.dctx, kpm, causal, S, Lq, Lk, H, d / H, drop_p, seed, site0, sc.dq, d, sc.dkv, sc.dkv + d, 2 * d, stream));
  TRY(linear_bwd(x, p->in_proj_w, sc.dq, Mq, d, d, dx, sc.ds, G(g->in_proj_w), G(g->in_proj_b), dws, sc, s, &defer));  // dx = dq·Wq + ds
  TRY(linear_bwd(c, p->in_proj_w + d * d, sc.dkv, Mk, 2 * d, d, dc, nullptr, g->in_proj_w ? G(g->in_proj_w) + d * d : nullptr,
                 g->in_proj_b ? G(g->in_proj_b) + d : nullptr, dws, sc, s, &defer));
  TRY(defer.flush());
  return flush_dw(dws, sc, s);
}

int stlt_ffn_block_fwd_train(const stlt_ffn_block_params* p, int64_t d, float eps, int act, int inner_dropout, const float* x, int64_t M,
                             float drop_p, uint64_t seed, uint32_t site0, float* u, float* h, float* f, float* out, void* gemm_scratch,
                             size_t gemm_scratch_bytes, stlt_stream_t stream) {
  if (!p || !x || !h || !f || !out || (act == STLT_ACT_GELU && !u)) return stlt_set_error(STLT_EINVAL, "stlt_ffn_block_fwd_train: null argument");
  if (gemm_scratch && gemm_scratch_bytes < STLT_GEMM_SCRATCH_BYTES) return stlt_set_error(STLT_EWORKSPACE, "stlt_ffn_block_fwd_train: gemm scratch too small");
  StltGemmScratch lend(gemm_scratch, STLT_GEMM_SCRATCH_BYTES);
  if (act != STLT_ACT_GELU && act != STLT_ACT_RELU) return stlt_set_error(STLT_EINVAL, "stlt_ffn_block_fwd_train: activation must be GELU or ReLU");
  if (!(drop_p >= 0.f && drop_p < 1.f)) return stlt_set_error(STLT_EINVAL, "dropout probability must be in [0,1)");
  hipStream_t s = (hipStream_t)stream;
  const StltDrop dr = stlt_drop_make(drop_p, seed);
  const StltDrop inner = inner_dropout ? dr : StltDrop{0u, 1.0f, 0ull};
  if (act == STLT_ACT_GELU) {
    TRY(launch_linear_gelu_keep(x, d, p->lin1_w, p->lin1_b, u, h, M, 4 * d, d, inner, site0, nullptr, s));
  } else {
    TRY(launch_linear(x, d, p->lin1_w, p->lin1_b, h, 4 * d, M, 4 * d, d, STLT_ACT_RELU, s));
    if (inner.thr) TRY(stlt_dropout(h, h, M * 4 * d, drop_p, seed, site0, stream));
  }
  TRY(launch_linear(h, 4 * d, p->lin2_w, p->lin2_b, f, d, M, d, 4 * d, STLT_ACT_NONE, s));
  return launch_add_layernorm(f, d, x, d, p->ln_w, p->ln_b, eps, M, d, out, d, s, dr, site0 + 1);
}

int stlt_ffn_block_bwd_train(const stlt_ffn_block_params* p, const stlt_ffn_block_params* g, int64_t d, float eps, int act, int inner_dropout,
                             const float* x, int64_t M, float drop_p, uint64_t seed, uint32_t site0, const float* u, const float* h, const float* f,
                             const float* dy, float* dx, stlt_ctx* tctx, void* keep, size_t keep_bytes, void* work, size_t work_bytes, stlt_stream_t stream) {
  if (!p || !g || !x || !h || !f || !dy || !dx || !keep || !work || (act == STLT_ACT_GELU && !u)) return stlt_set_error(STLT_EINVAL, "stlt_ffn_block_bwd_train: null argument");
  if (M < 0 || M > 0x7fffffff || d <= 0 || d % 4 != 0 || d > 16384) return stlt_set_error(STLT_EINVAL, "stlt_ffn_block_bwd_train: bad shape (M %lld, d %lld)", (long long)M, (long long)d);
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const BlockScratch sc = block_scratch((char*)keep, (char*)work, M, d, BLOCK_FFN);
  if (keep_bytes < sc.keep_total) return stlt_set_error(STLT_EWORKSPACE, "stlt_ffn_block_bwd_train: keep buffer %zu B < required %zu B", keep_bytes, sc.keep_total);
  if (work_bytes < sc.work_total) return stlt_set_error(STLT_EWORKSPACE, "stlt_ffn_block_bwd_train: work buffer %zu B < required %zu B", work_bytes, sc.work_total);
  StltCtxScope scope(tctx, s);
  if (scope.error()) return scope.error();
  const StltDrop dr = stlt_drop_make(drop_p, seed);
  const StltDrop inner = inner_dropout ? dr : StltDrop{0u, 1.0f, 0ull};
  auto G = [&](const float* q_) { return const_cast<float*>(q_); };
  BlockDefer defer(sc, s);
  float* df = dr.thr ? sc.da : sc.ds;
  TRY(launch_ln_bwd(dy, d, x, d, f, d, p->ln_w, eps, M, d, sc.ds, d, G(g->ln_w), G(g->ln_b), defer.chunk(sc), s, dr, site0 + 1, sc.da, 0, G(g->lin2_b)));
  DwList dws;
  bool gelu_fused = false;
  if (act == STLT_ACT_GELU && g->lin1_b && (size_t)((M + 255) / 256 * 16 * 4 * d) <= sc.red_floats) {
    // du = drop(df·W2) ∘ gelu'(u) in the dX product's epilogue + the bias column sums (train.hip's fused form); lin2_w += dfᵀ·h queued
    StltGemmScratch lend(sc.lin, STLT_GEMM_SCRATCH_BYTES);
    TRY(stlt_ffn_hidden_backward_fused(df, p->lin2_w, u, sc.dh, M, d, G(g->lin1_b), defer.chunk(sc), inner, site0, s, &gelu_fused));
    if (gelu_fused) dws.add(df, d, h, 4 * d, M, G(g->lin2_w));
  }
  if (!gelu_fused) TRY(linear_bwd(h, p->lin2_w, df, M, d, 4 * d, sc.dh, nullptr, G(g->lin2_w), nullptr, dws, sc, s));  // dh = df·W2 ; lin2_w += dfᵀ·h
  if (gelu_fused) {
  } else if (act == STLT_ACT_GELU) {
    if (g->lin1_b) TRY(launch_gelu_bwd_colsum(sc.dh, u, sc.dh, M, 4 * d, G(g->lin1_b), defer.chunk(sc), s, inner, site0));
    else TRY(launch_gelu_bwd(sc.dh, u, sc.dh, M * 4 * d, s, inner, site0));
  } else {
    if (inner.thr) TRY(stlt_dropout(sc.dh, sc.dh, M * 4 * d, drop_p, seed, site0, stream));
    TRY(stlt_relu_bwd(sc.dh, h, sc.dh, M * 4 * d, stream));  // h > 0 <=> pre-activation > 0 and kept
    if (g->lin1_b) TRY(launch_colsum_acc(sc.dh, 4 * d, M, 4 * d, G(g->lin1_b), defer.chunk(sc), s));
  }
  TRY(linear_bwd(x, p->lin1_w, sc.dh, M, 4 * d, d, dx, sc.ds, G(g->lin1_w), nullptr, dws, sc, s));  // dx = du·W1 + ds ; lin1_w += duᵀ·x
  TRY(defer.flush());
  return flush_dw(dws, sc, s);
}

}  // extern "C"
