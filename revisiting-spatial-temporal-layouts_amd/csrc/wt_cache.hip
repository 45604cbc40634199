// Training-loop contexts (include/stlt_hip.h: stlt_ctx_create / _destroy) and the transposed weight copies they hold for a step's
// input-gradient products.  dX = dY·W reads W (n_out, k_in) "as it lies" on the small-tile kernel's WKN build: its [k][n] operand image is
// gathered with four-byte LDS reads and runs 13 - 17 % below the forward build on the same shape (profiles/round5_gemm16_shapes.txt: qkv_dx
// against in_dx, ffn1_dx against ffn2, ...).  With a copy Wt (k_in, n_out) the same product is a forward product  dX = dY·(Wt)ᵀ  on the
// forward build, residual add-source and GELU-backward epilogue included.  The weights change once per optimisation step, so the trainer
// refreshes every copy at the start of its step (stlt_ctx_wt_refresh: batched 64 x 64 LDS transposes, ~1.2 GB of traffic for CACNF's 150 M
// weight elements: 0.25 ms) and withdraws them at its end (stlt_ctx_wt_clear).  The set belongs to ONE context: only calls that name that
// context can be served from it, on the device it was refreshed on, and every such call's stream is ordered behind the transposes (an event
// recorded at the end of the refresh) before it launches anything.  Lookup is by weight address — interior pointers included: the
// cross-attention blocks pass row ranges of in_proj_weight (models.py:362-382); a row range must start at a multiple of four rows (the copy's
// operand base is read with 16-byte loads).
#include <new>
#include "ctx.h"

namespace {

thread_local stlt_ctx* t_ctx = nullptr;

constexpr int WT_BATCH = 32;
struct WtBatch {
  int n;
  const float* w[WT_BATCH];
  float* wt[WT_BATCH];
  int rows[WT_BATCH], cols[WT_BATCH];  // w is rows x cols (n_out x k_in), wt cols x rows
  int tile0[WT_BATCH + 1];             // first 64 x 64 tile of every matrix in the launch's grid
};

__global__ __launch_bounds__(256) void wt_transpose_kernel(const WtBatch b) {
  __shared__ float t[64][65];
  int m = 0;
  while (m + 1 < b.n && (int)blockIdx.x >= b.tile0[m + 1]) ++m;
  const int rows = b.rows[m], cols = b.cols[m];
  const int tiles_x = (cols + 63) / 64;
  const int tile = blockIdx.x - b.tile0[m];
  const int r0 = (tile / tiles_x) * 64, c0 = (tile % tiles_x) * 64;
  const float* __restrict__ src = b.w[m];
  float* __restrict__ dst = b.wt[m];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 float4 columns x 16 rows per pass
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 16 * i, c = c0 + 4 * tx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows && c < cols) v = *reinterpret_cast<const f32x4*>(src + (int64_t)r * cols + c);  // cols % 4 == 0: a group is inside or outside as a whole
    t[ty + 16 * i][4 * tx + 0] = v[0];
    t[ty + 16 * i][4 * tx + 1] = v[1];
    t[ty + 16 * i][4 * tx + 2] = v[2];
    t[ty + 16 * i][4 * tx + 3] = v[3];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 16 * i, r = r0 + 4 * tx;  // output row c (a column of w), four consecutive input rows
    if (c < cols && r < rows) {  // rows % 4 == 0
      const f32x4 v = {t[4 * tx + 0][ty + 16 * i], t[4 * tx + 1][ty + 16 * i], t[4 * tx + 2][ty + 16 * i], t[4 * tx + 3][ty + 16 * i]};
      *reinterpret_cast<f32x4*>(dst + (int64_t)c * rows + r) = v;
    }
  }
}

}  // namespace

stlt_ctx* stlt_ctx_current() { return t_ctx; }

StltCtxScope::StltCtxScope(stlt_ctx* c, hipStream_t s) : prev_(t_ctx) {
  if (c && !stlt_ctx_valid(c)) { err_ = stlt_set_error(STLT_EINVAL, "stlt_ctx: not a live context handle"); c = nullptr; }
  t_ctx = c;
  if (!c) return;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->wt.empty() || s == c->wt_stream || c->wt_device != stlt_current_device()) return;
  if (!c->wt_wait_all)
    for (int i = 0; i < c->wt_n_waited; ++i) if (c->wt_waited[i] == s) return;
  if (hipError_t e = hipStreamWaitEvent(s, c->wt_ready, 0); e != hipSuccess) { err_ = stlt_set_error((int)e, "stlt_ctx: %s", hipGetErrorString(e)); return; }
  if (c->wt_n_waited < 4) c->wt_waited[c->wt_n_waited++] = s; else c->wt_wait_all = true;
}
StltCtxScope::~StltCtxScope() { t_ctx = prev_; }

// The current copy of rows [w, w + n_out*k_in) of a weight registered in the calling thread's current context: *wt points at the copy's first
// column of that row range, *ldwt is the copy's row pitch (the whole weight's n_out).  false: no context, nothing current in it (outside a
// trainer step), another device, a weight that was not registered, or a row range whose copy would not be 16-byte aligned.
bool stlt_wt_lookup(const float* w, int64_t n_out, int64_t k_in, const float** wt, int64_t* ldwt) {
  stlt_ctx* c = t_ctx;
  if (!c) return false;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->wt.empty() || c->wt_device != stlt_current_device()) return false;
  auto it = c->wt.upper_bound((uintptr_t)w);
  if (it == c->wt.begin()) return false;
  --it;
  const StltWtEnt& e = it->second;
  if (e.k_in != k_in || w < e.w) return false;
  const int64_t off = w - e.w;
  if (off % k_in != 0) return false;
  const int64_t r0 = off / k_in;
  if (r0 + n_out > e.n_out || r0 % 4 != 0) return false;
  *wt = e.wt + r0;
  *ldwt = e.n_out;
  return true;
}
void stlt_wt_count_hit() {
  if (stlt_ctx* c = t_ctx) { std::lock_guard<std::mutex> lk(c->mu); ++c->wt_hits; }
}

extern "C" {

int stlt_ctx_create(stlt_ctx** out) {
  if (!out) return stlt_set_error(STLT_EINVAL, "stlt_ctx_create: null argument");
  stlt_ctx* c = new (std::nothrow) stlt_ctx();
  if (!c) return stlt_set_error(STLT_EINVAL, "stlt_ctx_create: out of host memory");
  *out = c;
  return 0;
}

// The caller has synchronised with everything it enqueued through the context (its streams are idle or will not touch the context's side
// streams again): the side streams and events are destroyed, queued weight gradients are dropped, the copies are withdrawn.
int stlt_ctx_destroy(stlt_ctx* c) {
  if (!c) return 0;
  if (!stlt_ctx_valid(c)) return stlt_set_error(STLT_EINVAL, "stlt_ctx_destroy: not a live context handle");
  if (t_ctx == c) t_ctx = nullptr;
  for (int d = 0; d < STLT_MAX_DEVICES; ++d) {
    StltSideDevice& dv = c->side[d];
    if (!dv.tried) continue;
    for (int i = 0; i < 4; ++i) if (dv.ev[i]) (void)hipEventDestroy(dv.ev[i]);
    if (dv.s) { (void)hipStreamSynchronize(dv.s); (void)hipStreamDestroy(dv.s); }
  }
  if (c->wt_ready) (void)hipEventDestroy(c->wt_ready);
  (void)hipGetLastError();
  c->magic = 0;
  delete c;
  return 0;
}

int stlt_ctx_wt_refresh(stlt_ctx* c, const stlt_wt_entry* entries, int64_t n, stlt_stream_t stream) {
  if (!stlt_ctx_valid(c)) return stlt_set_error(STLT_EINVAL, "stlt_ctx_wt_refresh: not a live context handle");
  if (n < 0 || (n > 0 && !entries)) return stlt_set_error(STLT_EINVAL, "stlt_ctx_wt_refresh: null table");
  hipStream_t s = (hipStream_t)stream;
  for (int64_t i = 0; i < n; ++i) {
    const stlt_wt_entry& e = entries[i];
    if (!e.w || !e.wt || e.n_out <= 0 || e.k_in <= 0 || e.n_out % 4 || e.k_in % 4 || e.n_out > 0x3fffff || e.k_in > 0x3fffff ||
        (((uintptr_t)e.w | (uintptr_t)e.wt) & 15))
      return stlt_set_error(STLT_EINVAL, "stlt_ctx_wt_refresh: entry %lld: 16-byte aligned pointers and dimensions that are multiples of 4 are required", (long long)i);
  }
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->wt.clear();  // nothing is current while the copies are being rewritten
    if (c->wt_ready && c->wt_device != stlt_current_device()) {  // an event belongs to the device it was created on
      (void)hipEventDestroy(c->wt_ready);
      (void)hipGetLastError();
      c->wt_ready = nullptr;
    }
    if (!c->wt_ready && hipEventCreateWithFlags(&c->wt_ready, hipEventDisableTiming) != hipSuccess) {
      c->wt_ready = nullptr;
      (void)hipGetLastError();
      return stlt_set_error(STLT_EINVAL, "stlt_ctx_wt_refresh: could not create the ready event");
    }
  }
  for (int64_t i0 = 0; i0 < n; i0 += WT_BATCH) {
    WtBatch b;
    b.n = (int)((n - i0) < WT_BATCH ? (n - i0) : WT_BATCH);
    int64_t tiles = 0;
    for (int j = 0; j < b.n; ++j) {
      const stlt_wt_entry& e = entries[i0 + j];
      b.w[j] = e.w; b.wt[j] = e.wt; b.rows[j] = (int)e.n_out; b.cols[j] = (int)e.k_in;
      b.tile0[j] = (int)tiles;
      tiles += ((e.n_out + 63) / 64) * ((e.k_in + 63) / 64);
      if (tiles > 0x3fffffffLL) return stlt_set_error(STLT_EINVAL, "stlt_ctx_wt_refresh: too many tiles");
    }
    b.tile0[b.n] = (int)tiles;
    for (int j = b.n + 1; j <= WT_BATCH; ++j) b.tile0[j] = (int)tiles;
    hipLaunchKernelGGL(wt_transpose_kernel, dim3((unsigned)tiles), dim3(256), 0, s, b);
    if (int e = stlt_check_launch("wt_transpose_kernel")) return e;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  if (n > 0) {
    if (hipError_t e = hipEventRecord(c->wt_ready, s); e != hipSuccess) return stlt_set_error((int)e, "stlt_ctx_wt_refresh: %s", hipGetErrorString(e));
  }
  c->wt_device = stlt_current_device();
  c->wt_stream = s;
  c->wt_n_waited = 0;
  c->wt_wait_all = false;
  for (int64_t i = 0; i < n; ++i) c->wt[(uintptr_t)entries[i].w] = StltWtEnt{entries[i].w, entries[i].wt, entries[i].n_out, entries[i].k_in};
  return 0;
}

int stlt_ctx_wt_clear(stlt_ctx* c) {
  if (!stlt_ctx_valid(c)) return stlt_set_error(STLT_EINVAL, "stlt_ctx_wt_clear: not a live context handle");
  std::lock_guard<std::mutex> lk(c->mu);
  c->wt.clear();
  return 0;
}

long long stlt_ctx_wt_hits(stlt_ctx* c) {
  if (!stlt_ctx_valid(c)) return -1;
  std::lock_guard<std::mutex> lk(c->mu);
  return c->wt_hits;
}

}  // extern "C"
