// Ragged layout of a padded batch: which (clip, frame, object) tokens are real, and where they go when only the
// real ones are kept.
//
// The collater (reference datasets.py:239-288) pads every clip to T frames and every frame to N object slots; the
// reference then runs both transformers over all B*T*N / B*T rows, padded ones included.  A padded row never
// influences a real one (it is masked as a key, and no real query ever reads it), and Stlt.forward only reads one
// real row per clip, so the real rows can be computed on their own: GEMM and LayerNorm rows are independent, and an
// attention row over its real keys is the same sum with the masked (zero-weight) terms left out.
//
// Order is preserved: compacted tokens are sorted by (clip, frame, slot), compacted frames by (clip, frame).  A
// frame's tokens form one spatial segment; a clip's frames one temporal segment.
#include "common.h"

namespace {

// real tokens / frames of each clip.  One block per clip, thread t = frame t (T <= 256).
__global__ __launch_bounds__(256) void ragged_count_kernel(const uint8_t* __restrict__ kpm_boxes,
                                                           const uint8_t* __restrict__ kpm_frames, int T, int N,
                                                           int* __restrict__ clip_tok, int* __restrict__ clip_frm, int* __restrict__ counts) {
  __shared__ int s_tok[256], s_frm[256];
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;
  // the contract flag starts from zero for this index (the fill kernel, two launches later, only ever raises it).  Cleared here and not by
  // a memset: a memset node of a captured graph was seen to run out of order with the kernels around it on replay (round 6: the second
  // replay of a skip-padding forward found the freshly written counts zeroed)
  if (b == 0 && t == 0) { counts[2] = 0; counts[3] = 0; }
  int cnt = 0, real = 0;
  if (t < T && kpm_frames[b * T + t] == 0) {
    real = 1;
    const uint8_t* m = kpm_boxes + (b * T + t) * N;
    for (int n = 0; n < N; ++n) cnt += m[n] == 0;
  }
  s_tok[t] = cnt;
  s_frm[t] = real;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { s_tok[t] += s_tok[t + o]; s_frm[t] += s_frm[t + o]; }
    __syncthreads();
  }
  if (t == 0) { clip_tok[b] = s_tok[0]; clip_frm[b] = s_frm[0]; }
}

// exclusive prefix sums over the clips (one block; B is at most a few thousand)
__global__ __launch_bounds__(1024) void ragged_scan_kernel(const int* __restrict__ clip_tok, const int* __restrict__ clip_frm,
                                                           int64_t B, int* __restrict__ tok_off, int* __restrict__ frm_off,
                                                           int* __restrict__ counts) {
  __shared__ int s_a[1024], s_b[1024];
  __shared__ int carry_a, carry_b;
  const int t = threadIdx.x;
  if (t == 0) { carry_a = 0; carry_b = 0; }
  __syncthreads();
  for (int64_t base = 0; base < B; base += 1024) {
    const int64_t i = base + t;
    const int a = i < B ? clip_tok[i] : 0, c = i < B ? clip_frm[i] : 0;
    s_a[t] = a; s_b[t] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
      const int va = t >= o ? s_a[t - o] : 0, vb = t >= o ? s_b[t - o] : 0;
      __syncthreads();
      s_a[t] += va; s_b[t] += vb;
      __syncthreads();
    }
    if (i < B) { tok_off[i] = carry_a + s_a[t] - a; frm_off[i] = carry_b + s_b[t] - c; }
    __syncthreads();
    if (t == 1023) { carry_a += s_a[1023]; carry_b += s_b[1023]; }
    __syncthreads();
  }
  if (t == 0) { counts[0] = carry_a; counts[1] = carry_b; frm_off[B] = carry_b; }
}

// per clip: place its real frames / tokens.  thread t = frame t.
__global__ __launch_bounds__(256) void ragged_fill_kernel(const uint8_t* __restrict__ kpm_boxes,
                                                          const uint8_t* __restrict__ kpm_frames,
                                                          const int64_t* __restrict__ lengths, int T, int N, RaggedIndex ix) {
  __shared__ int s_tok[256], s_frm[256];
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;
  int cnt = 0, real = 0;
  const uint8_t* m = kpm_boxes + (b * T + (t < T ? t : 0)) * N;
  if (t < T && kpm_frames[b * T + t] == 0) {
    real = 1;
    for (int n = 0; n < N; ++n) cnt += m[n] == 0;
  }
  s_tok[t] = cnt;
  s_frm[t] = real;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int va = t >= o ? s_tok[t - o] : 0, vb = t >= o ? s_frm[t - o] : 0;
    __syncthreads();
    s_tok[t] += va; s_frm[t] += vb;
    __syncthreads();
  }
  const int frm0 = ix.clip_frm_off[b], n_frm = ix.clip_frm[b];
  const int row0 = ix.clip_tok_off[b] + s_tok[t] - cnt;  // first token row of this frame
  const int fr = frm0 + s_frm[t] - real;                  // this frame's row
  if (t < T) ix.f_row_of[b * T + t] = real ? fr : -1;
  if (real) {
    ix.f_orig[fr] = (int)(b * T + t);
    ix.f_cls_row[fr] = row0;
    ix.f_seg_start[fr] = frm0;
    ix.f_seg_end[fr] = frm0 + n_frm;
    if (m[0] != 0) ix.counts[2] = 1;  // slot 0 must be the (real) CLS object (datasets.py:247-264)
    int r = row0;
    for (int n = 0; n < N; ++n) {
      if (m[n] == 0) {
        ix.t_orig[r] = (int)((b * T + t) * N + n);
        ix.t_seg_start[r] = row0;
        ix.t_seg_end[r] = row0 + cnt;
        ++r;
      }
    }
  }
  if (lengths) {
    int64_t last = lengths[b] - 1;
    last = last < 0 ? last + T : last;  // python indexing of lengths-1 == -1
    if (t == last) {
      if (real) ix.last_row[b] = fr;
      else { ix.last_row[b] = 0; ix.counts[2] = 1; }  // the row the head reads must be a real frame
    }
    if (t == 0 && (last < 0 || last >= T)) { ix.last_row[b] = 0; ix.counts[2] = 1; }
  }
}

// padded schedule: the rows the tail layers pick — CLS token of every frame, frame lengths-1 of every clip
__global__ __launch_bounds__(256) void padded_rows_kernel(const int64_t* __restrict__ lengths, int64_t B, int T, int N,
                                                          int* __restrict__ cls_row, int* __restrict__ last_row) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < B * T) cls_row[i] = (int)(i * N);
  if (i < B) {
    int64_t t = lengths[i] - 1;
    t = t < 0 ? t + T : t;  // python indexing of lengths-1 == -1
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    last_row[i] = (int)(i * T + t);
  }
}

__global__ __launch_bounds__(256) void ragged_groups_kernel(const int* __restrict__ f_cls_row, int n_tokens, int n_frames, int fpg,
                                                            int n_groups, int* __restrict__ grp_ptr) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g > n_groups) return;
  const int64_t f = (int64_t)g * fpg;
  grp_ptr[g] = f < n_frames ? f_cls_row[f] : n_tokens;
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ rows, int64_t n, int d,
                                                           float* __restrict__ dst, const int* __restrict__ n_dev) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n || (n_dev && i >= *n_dev)) return;  // n_dev: the real row count on the device, when n is only an upper bound (rows beyond it are dummies)
  float* o = dst + (int64_t)rows[i] * d;
  for (int e = lane * 4; e < d; e += 256) *reinterpret_cast<f32x4*>(o + e) = *reinterpret_cast<const f32x4*>(src + i * d + e);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t ld, const int* __restrict__ rows,
                                                          int64_t n, int d, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* s = src + (int64_t)rows[i] * ld;
  for (int e = lane * 4; e < d; e += 256) *reinterpret_cast<f32x4*>(out + i * d + e) = *reinterpret_cast<const f32x4*>(s + e);
}

inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

// Row counts given by the caller (stlt_inputs.n_real_tokens / n_real_frames: a collater knows them on the host) instead of read back.  Rows
// between the index's own counts and the caller's get a self-contained entry (token 0 / frame 0, a one-row segment), so that a count that is
// too LARGE stays inside every buffer; stlt_ragged_poison then turns the call's result into NaN when the caller's counts are not the index's
// (or the masks break the collater contract) — the check the read-back made on the host, moved behind the launches it used to precede.
__global__ __launch_bounds__(256) void ragged_host_counts_kernel(RaggedIndex ix, int n_tok, int n_frm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int have_tok = ix.counts[0], have_frm = ix.counts[1];
  if (i >= have_tok && i < n_tok) { ix.t_orig[i] = 0; ix.t_seg_start[i] = i; ix.t_seg_end[i] = i + 1; }
  if (i >= have_frm && i < n_frm) { ix.f_orig[i] = 0; ix.f_cls_row[i] = 0; ix.f_seg_start[i] = i; ix.f_seg_end[i] = i + 1; }
}
// dst = 0 as a kernel (not hipMemsetAsync: see ragged_count_kernel — a captured graph must replay this before the scatter that follows)
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) reinterpret_cast<f32x4*>(dst)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// allow_more: the caller's counts are upper bounds (inference: the rows between the real counts and the bounds are self-contained dummy rows
// nobody reads); else they must be the index's own (training: a dummy row would leave a gradient)
__global__ __launch_bounds__(256) void ragged_poison_kernel(const int* __restrict__ counts, int n_tok, int n_frm, int allow_more, float* __restrict__ out, int64_t n) {
  const bool ok = allow_more ? (counts[0] <= n_tok && counts[1] <= n_frm) : (counts[0] == n_tok && counts[1] == n_frm);
  if (ok && counts[2] == 0) return;
  const float nan = __int_as_float(0x7fc00000);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = nan;
}

}  // namespace

int launch_ragged_host_counts(const RaggedIndex& ix, int64_t n_tok, int64_t n_frm, int64_t max_tok, int64_t max_frm, hipStream_t s) {
  if (n_tok <= 0 || n_frm <= 0 || n_tok > max_tok || n_frm > max_frm || n_frm > n_tok)
    return stlt_set_error(STLT_EINVAL, "skip-padding: n_real_tokens = %lld / n_real_frames = %lld do not fit the batch (%lld tokens, %lld frames)", (long long)n_tok,
                          (long long)n_frm, (long long)max_tok, (long long)max_frm);
  StltProfScope ps(STLT_K_MISC, s);
  hipLaunchKernelGGL(ragged_host_counts_kernel, dim3((unsigned)((n_tok + 255) / 256)), dim3(256), 0, s, ix, (int)n_tok, (int)n_frm);
  return stlt_check_launch("ragged_host_counts_kernel");
}
int launch_ragged_poison(const RaggedIndex& ix, int64_t n_tok, int64_t n_frm, bool allow_more, float* out, int64_t n, hipStream_t s) {
  if (!out || n <= 0) return 0;
  StltProfScope ps(STLT_K_MISC, s);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(ragged_poison_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ix.counts, (int)n_tok, (int)n_frm, allow_more ? 1 : 0, out, n);
  return stlt_check_launch("ragged_poison_kernel");
}

size_t ragged_index_bytes(int64_t B, int64_t T, int64_t N) {
  const size_t tok = (size_t)B * T * N, bt = (size_t)B * T;
  return 3 * al(tok * 4) + 4 * al(bt * 4) + 4 * al((size_t)B * 4) + al((size_t)(B + 1) * 4) + al(bt * 4) + al((bt + 1) * 4) + 256;
}

RaggedIndex ragged_index_carve(void* base, int64_t B, int64_t T, int64_t N) {
  const size_t tok = (size_t)B * T * N, bt = (size_t)B * T;
  char* p = (char*)base;
  auto take = [&](size_t n) { int* q = (int*)p; p += al(n * 4); return q; };
  RaggedIndex ix;
  ix.t_seg_start = take(tok); ix.t_seg_end = take(tok); ix.t_orig = take(tok);
  ix.f_seg_start = take(bt); ix.f_seg_end = take(bt); ix.f_orig = take(bt); ix.f_cls_row = take(bt);
  ix.last_row = take(B); ix.clip_tok = take(B); ix.clip_frm = take(B); ix.clip_tok_off = take(B); ix.clip_frm_off = take(B + 1);
  ix.f_row_of = take(bt); ix.sp_grp_ptr = take(bt + 1);
  ix.counts = take(4);
  return ix;
}

int launch_ragged_index(const uint8_t* kpm_boxes, const uint8_t* kpm_frames, const int64_t* lengths, int64_t B, int64_t T,
                        int64_t N, const RaggedIndex& ix, hipStream_t s) {
  StltProfScope ps(STLT_K_MISC, s);
  if (!kpm_boxes || !kpm_frames) return stlt_set_error(STLT_EINVAL, "ragged_index: null mask");
  if (T > 256) return stlt_set_error(STLT_EINVAL, "ragged_index: T=%lld > 256", (long long)T);
  if (B * T * N > 0x7fffff00LL) return stlt_set_error(STLT_EINVAL, "ragged_index: batch too large for 32-bit row indices");
  hipLaunchKernelGGL(ragged_count_kernel, dim3((unsigned)B), dim3(256), 0, s, kpm_boxes, kpm_frames, (int)T, (int)N, ix.clip_tok, ix.clip_frm, ix.counts);
  if (int e = stlt_check_launch("ragged_count_kernel")) return e;
  hipLaunchKernelGGL(ragged_scan_kernel, dim3(1), dim3(1024), 0, s, ix.clip_tok, ix.clip_frm, B, ix.clip_tok_off, ix.clip_frm_off, ix.counts);
  if (int e = stlt_check_launch("ragged_scan_kernel")) return e;
  hipLaunchKernelGGL(ragged_fill_kernel, dim3((unsigned)B), dim3(256), 0, s, kpm_boxes, kpm_frames, lengths, (int)T, (int)N, ix);
  return stlt_check_launch("ragged_fill_kernel");
}

int launch_gather_rows(const float* src, int64_t ld, const int* rows, int64_t n, int64_t d, float* out, hipStream_t s) {
  if (!src || !rows || !out) return stlt_set_error(STLT_EINVAL, "gather_rows: null pointer");
  if (d % 4 || ld % 4) return stlt_set_error(STLT_EINVAL, "gather_rows: d and ld must be multiples of 4");
  if (n == 0) return 0;
  StltProfScope ps(STLT_K_GATHER, s);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, src, ld, rows, n, (int)d, out);
  return stlt_check_launch("gather_rows_kernel");
}

int launch_ragged_groups(const RaggedIndex& ix, int64_t n_tokens, int64_t n_frames, int fpg, hipStream_t s) {
  StltProfScope ps(STLT_K_MISC, s);
  if (fpg < 1) return stlt_set_error(STLT_EINVAL, "ragged_groups: frames per group must be positive");
  const int64_t n_groups = (n_frames + fpg - 1) / fpg;
  hipLaunchKernelGGL(ragged_groups_kernel, dim3((unsigned)((n_groups + 256) / 256)), dim3(256), 0, s, ix.f_cls_row, (int)n_tokens,
                     (int)n_frames, fpg, (int)n_groups, ix.sp_grp_ptr);
  return stlt_check_launch("ragged_groups_kernel");
}

int launch_scatter_rows(const float* src, const int* rows, int64_t n, int64_t d, float* dst, int64_t dst_rows, hipStream_t s, const int* n_dev) {
  StltProfScope ps(STLT_K_MISC, s);
  if (!src || !rows || !dst) return stlt_set_error(STLT_EINVAL, "scatter_rows: null pointer");
  if (d % 4) return stlt_set_error(STLT_EINVAL, "scatter_rows: d must be a multiple of 4");
  if (dst_rows > 0) {
    if (((uintptr_t)dst & 15) != 0) return stlt_set_error(STLT_EINVAL, "scatter_rows: dst must be 16-byte aligned");
    const int64_t n4 = dst_rows * d / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, n4);
    if (int e = stlt_check_launch("zero_rows_kernel")) return e;
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, src, rows, n, (int)d, dst, n_dev);
  return stlt_check_launch("scatter_rows_kernel");
}

int launch_padded_rows(const int64_t* lengths, int64_t B, int64_t T, int64_t N, const RaggedIndex& ix, hipStream_t s) {
  StltProfScope ps(STLT_K_MISC, s);
  if (!lengths) return stlt_set_error(STLT_EINVAL, "padded_rows: null lengths");
  if (B * T * N > 0x7fffff00LL) return stlt_set_error(STLT_EINVAL, "padded_rows: batch too large for 32-bit row indices");
  hipLaunchKernelGGL(padded_rows_kernel, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, s, lengths, B, (int)T, (int)N, ix.f_cls_row, ix.last_row);
  return stlt_check_launch("padded_rows_kernel");
}
