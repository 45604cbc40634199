// Row-wise HBM-bound kernels: K1 fused category+box(+score) embedding + LayerNorm, residual+LayerNorm,
// K7 frames embedding (CLS select + position + frame-type + LayerNorm), K8a last-state gather.
//
// One 64-lane wavefront owns one row of d floats, held entirely in registers (NV float4 per lane,
// 16-byte coalesced loads/stores), mean and variance by wave shuffles: no LDS, no second HBM pass.
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves / 256 threads

template <int NV>
__device__ __forceinline__ void ln_store(f32x4 (&v)[NV], int lane, int d, const float* __restrict__ w,
                                         const float* __restrict__ b, float eps, float* __restrict__ out,
                                         const StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0, uint64_t row_idx0 = 0) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float inv_d = 1.0f / (float)d;
  const float mean = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) {
      v[i] -= mean;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float var = wave_sum(q) * inv_d;  // biased, as nn.LayerNorm
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) {
      f32x4 ww = *reinterpret_cast<const f32x4*>(w + e);
      f32x4 bb = *reinterpret_cast<const f32x4*>(b + e);
      f32x4 o = v[i] * rstd * ww + bb;
      if (dr.thr) o = stlt_drop4(dr, site, row_idx0 + e, o);  // train-mode dropout on the LayerNorm output
      *reinterpret_cast<f32x4*>(out + e) = o;
    }
  }
}

// ---------------------------------------------------------------- K1
template <int NV>
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ categories,
                                                    const float* __restrict__ boxes,
                                                    const float* __restrict__ scores,
                                                    const float* __restrict__ cat_table, int n_categories,
                                                    const float* __restrict__ box_w, const float* __restrict__ box_b,
                                                    const float* __restrict__ score_w,
                                                    const float* __restrict__ score_b, const float* __restrict__ ln_w,
                                                    const float* __restrict__ ln_b, float eps, int64_t n_tokens, int d,
                                                    float* __restrict__ out, float* __restrict__ pre_out, StltDrop dr,
                                                    const int* __restrict__ src_index) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= n_tokens) return;
  const int64_t src = src_index ? src_index[row] : row;  // ragged mode: output row -> token of the padded batch
  int64_t cat = categories[src];
  cat = cat < 0 ? 0 : (cat >= n_categories ? n_categories - 1 : cat);  // never read outside the table
  const f32x4 box = *reinterpret_cast<const f32x4*>(boxes + src * 4);
  const float sc = scores ? scores[src] : 0.f;
  const float* __restrict__ erow = cat_table + cat * d;
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) {
      f32x4 x = *reinterpret_cast<const f32x4*>(erow + e);
      f32x4 bb = *reinterpret_cast<const f32x4*>(box_b + e);
      f32x4 lin;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 wr = *reinterpret_cast<const f32x4*>(box_w + (int64_t)(e + c) * 4);  // box_w is (d,4)
        lin[c] = ((box.x * wr.x + box.y * wr.y) + (box.z * wr.z + box.w * wr.w)) + bb[c];
      }
      x += lin;
      if (scores) {
        f32x4 ws = *reinterpret_cast<const f32x4*>(score_w + e);  // (d,1)
        f32x4 bs = *reinterpret_cast<const f32x4*>(score_b + e);
        x += sc * ws + bs;
      }
      v[i] = x;
      if (pre_out) *reinterpret_cast<f32x4*>(pre_out + row * d + e) = x;  // training tape: pre-LayerNorm sum
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  ln_store<NV>(v, lane, d, ln_w, ln_b, eps, out + row * d, dr, STLT_SITE_EMBED, (uint64_t)row * d);
}

// ---------------------------------------------------------------- residual + LN
template <int NV>
__global__ __launch_bounds__(256) void add_ln_kernel(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ res, int64_t ldres,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     float eps, int64_t M, int d, float* __restrict__ out,
                                                     int64_t ldout, StltDrop dr, uint32_t site,
                                                     const int* __restrict__ drop_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  // rows picked out of a larger buffer keep the dropout masks of their original positions
  const uint64_t drow = (dr.thr && drop_rows) ? (uint64_t)drop_rows[row] : (uint64_t)row;
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) {
      v[i] = *reinterpret_cast<const f32x4*>(x + row * ldx + e);
      if (dr.thr) v[i] = stlt_drop4(dr, site, drow * d + e, v[i]);  // dropout1 / dropout2 of the encoder layer (before the residual)
      if (res) v[i] += *reinterpret_cast<const f32x4*>(res + row * ldres + e);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  ln_store<NV>(v, lane, d, w, b, eps, out + row * ldout);
}

// ---------------------------------------------------------------- K7
template <int NV>
__global__ __launch_bounds__(256) void frames_embed_kernel(const float* __restrict__ spatial, int64_t row_stride,
                                                           const int64_t* __restrict__ frame_types,
                                                           const float* __restrict__ pos_table,
                                                           const float* __restrict__ type_table,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           float eps, int64_t BT, int T, int d,
                                                           float* __restrict__ out, float* __restrict__ pre_out, StltDrop dr,
                                                           const int* __restrict__ src_index) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= BT) return;
  const int64_t src = src_index ? src_index[row] : row;  // ragged mode: output row -> frame b*T+t of the padded batch
  const int t = (int)(src % T);
  int64_t ft = frame_types[src];
  ft = ft < 0 ? 0 : (ft > 4 ? 4 : ft);  // frame_type_embedding has 5 rows (models.py:91)
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) {
      f32x4 a = *reinterpret_cast<const f32x4*>(spatial + row * row_stride + e);
      f32x4 p = *reinterpret_cast<const f32x4*>(pos_table + (int64_t)t * d + e);
      f32x4 f = *reinterpret_cast<const f32x4*>(type_table + ft * d + e);
      v[i] = (a + p) + f;  // models.py:108 evaluation order
      if (pre_out) *reinterpret_cast<f32x4*>(pre_out + row * d + e) = v[i];
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  ln_store<NV>(v, lane, d, w, b, eps, out + row * d, dr, STLT_SITE_FRAMES, (uint64_t)row * d);
}

// ---------------------------------------------------------------- K8a
__global__ __launch_bounds__(256) void gather_last_kernel(const float* __restrict__ x,
                                                          const int64_t* __restrict__ lengths, int64_t B, int T,
                                                          int d, float* __restrict__ out) {
  const int64_t b = blockIdx.x;
  int64_t t = lengths[b] - 1;
  t = t < 0 ? t + T : t;  // python negative indexing of lengths-1 == -1
  t = t < 0 ? 0 : (t >= T ? T - 1 : t);
  const float* src = x + (b * T + t) * (int64_t)d;
  for (int e = threadIdx.x * 4; e < d; e += blockDim.x * 4)
    *reinterpret_cast<f32x4*>(out + b * d + e) = *reinterpret_cast<const f32x4*>(src + e);
}

inline int nv_for(int64_t d) { return (int)((d + 255) / 256); }

#define DISPATCH_NV(nv, ...)                                   \
  switch (nv) {                                                \
    case 1: { constexpr int NV = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int NV = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int NV = 3; __VA_ARGS__; } break;      \
    case 4: { constexpr int NV = 4; __VA_ARGS__; } break;      \
    default: { constexpr int NV = 8; __VA_ARGS__; } break;     \
  }

inline int check_d(int64_t d) {
  if (d <= 0 || d % 4 != 0 || d > 2048) return stlt_set_error(STLT_EINVAL, "hidden size d=%lld must be a multiple of 4 in [4,2048]", (long long)d);
  return 0;
}

}  // namespace

int launch_embed(const int64_t* categories, const float* boxes, const float* scores, const float* cat_table,
                 int64_t n_categories, const float* box_w, const float* box_b, const float* score_w,
                 const float* score_b, const float* ln_w, const float* ln_b, float eps, int64_t n_tokens, int64_t d,
                 float* out, hipStream_t s, float* pre_out, StltDrop dr, const int* src_index) {
  if (int e = check_d(d)) return e;
  if (!categories || !boxes || !cat_table || !box_w || !box_b || !ln_w || !ln_b || !out || n_categories <= 0)
    return stlt_set_error(STLT_EINVAL, "stlt_embed_fwd: null pointer / empty table");
  if (scores && (!score_w || !score_b)) return stlt_set_error(STLT_EINVAL, "stlt_embed_fwd: scores given without score_w/score_b");
  if (n_tokens == 0) return 0;
  StltProfScope ps(STLT_K_EMBED, s);
  stlt_prof_note("embed rows=%lld d=%lld", (long long)n_tokens, (long long)d);
  stlt_prof_add_bytes((double)n_tokens * (4.0 * d * (pre_out ? 2 : 1) + 29.0));
  dim3 grid((unsigned)((n_tokens + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  DISPATCH_NV(nv_for(d), hipLaunchKernelGGL((embed_kernel<NV>), grid, dim3(256), 0, s, categories, boxes, scores,
                                            cat_table, (int)n_categories, box_w, box_b, score_w, score_b, ln_w, ln_b,
                                            eps, n_tokens, (int)d, out, pre_out, dr, src_index));
  return stlt_check_launch("embed_kernel");
}

int launch_add_layernorm(const float* x, int64_t ldx, const float* res, int64_t ldres, const float* w, const float* b,
                         float eps, int64_t M, int64_t d, float* out, int64_t ldout, hipStream_t s, StltDrop dr, uint32_t site,
                         const int* drop_rows) {
  if (int e = check_d(d)) return e;
  if (!x || !w || !b || !out) return stlt_set_error(STLT_EINVAL, "stlt_add_layernorm_fwd: null pointer");
  if (ldx % 4 || ldout % 4 || (res && ldres % 4)) return stlt_set_error(STLT_EINVAL, "stlt_add_layernorm_fwd: leading dims must be multiples of 4");
  if (M == 0) return 0;
  StltProfScope ps(STLT_K_ADDLN, s);
  stlt_prof_note("add_ln rows=%lld d=%lld%s", (long long)M, (long long)d, res ? " +res" : "");
  stlt_prof_add_bytes((double)M * 4.0 * d * (res ? 3 : 2));
  dim3 grid((unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  DISPATCH_NV(nv_for(d), hipLaunchKernelGGL((add_ln_kernel<NV>), grid, dim3(256), 0, s, x, ldx, res, ldres, w, b, eps,
                                            M, (int)d, out, ldout, dr, site, drop_rows));
  return stlt_check_launch("add_ln_kernel");
}

int launch_frames_embed(const float* spatial, int64_t row_stride, const int64_t* frame_types, const float* pos_table,
                        const float* type_table, const float* ln_w, const float* ln_b, float eps, int64_t B, int64_t T,
                        int64_t d, float* out, hipStream_t s, float* pre_out, StltDrop dr, const int* src_index, int64_t n_rows) {
  if (int e = check_d(d)) return e;
  if (!spatial || !frame_types || !pos_table || !type_table || !ln_w || !ln_b || !out)
    return stlt_set_error(STLT_EINVAL, "stlt_frames_embed_fwd: null pointer");
  if (row_stride % 4) return stlt_set_error(STLT_EINVAL, "stlt_frames_embed_fwd: row_stride must be a multiple of 4");
  const int64_t rows = src_index ? n_rows : B * T;  // ragged mode: n_rows compacted frames
  if (rows == 0) return 0;
  StltProfScope ps(STLT_K_FRAMES, s);
  stlt_prof_note("frames_embed rows=%lld d=%lld", (long long)rows, (long long)d);
  stlt_prof_add_bytes((double)rows * (4.0 * d * (pre_out ? 3 : 2) + 9.0));
  dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  DISPATCH_NV(nv_for(d), hipLaunchKernelGGL((frames_embed_kernel<NV>), grid, dim3(256), 0, s, spatial, row_stride,
                                            frame_types, pos_table, type_table, ln_w, ln_b, eps, rows, (int)T, (int)d,
                                            out, pre_out, dr, src_index));
  return stlt_check_launch("frames_embed_kernel");
}

int launch_gather_last(const float* x, const int64_t* lengths, int64_t B, int64_t T, int64_t d, float* out,
                       hipStream_t s) {
  if (int e = check_d(d)) return e;
  if (!x || !lengths || !out) return stlt_set_error(STLT_EINVAL, "stlt_gather_last_fwd: null pointer");
  if (B == 0) return 0;
  StltProfScope ps(STLT_K_GATHER, s);
  stlt_prof_note("gather_last rows=%lld d=%lld", (long long)B, (long long)d);
  stlt_prof_add_bytes((double)B * 8.0 * d);
  hipLaunchKernelGGL(gather_last_kernel, dim3((unsigned)B), dim3(256), 0, s, x, lengths, B, (int)T, (int)d, out);
  return stlt_check_launch("gather_last_kernel");
}
