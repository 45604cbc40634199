// Row-wise HBM-bound kernels: K1 fused category+box(+score) embedding + LayerNorm, residual+LayerNorm,
// K7 frames embedding (CLS select + position + frame-type + LayerNorm), K8a last-state gather.
//
// One 64-lane wavefront owns one row of d floats, held entirely in registers (NV float4 per lane,
// 16-byte coalesced loads/stores), mean and variance by wave shuffles: no LDS, no second HBM pass.
#include "common.h"
#include "wave_dpp.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves / 256 threads

// LayerNorm of a row held in registers + store; the scale / shift come from memory (LnParamsInMemory) or from registers (LnParamsInRegs:
// a wave that normalises several rows loads them once).  The products that may or may not be fused into multiply-adds are written out
// (contraction off, fmaf where one is wanted): every kernel of this file rounds a row the same way wherever this is inlined, so a row does
// not depend on which kernel (one token per wave, several tokens per wave) produced it.
struct LnParamsInMemory {
  // one row per wave at 8 waves / SIMD: the ds_bpermute butterfly's latency is covered by the other waves and its LDS steps run beside the
  // VALU; the DPP sum measured 3 % slower there (add_ln, 229 376 rows: 246.8 vs 239.8 us) — same bits either way
  static constexpr bool dpp_sums = false;
  const float* __restrict__ w;
  const float* __restrict__ b;
  __device__ __forceinline__ f32x4 scale(int, int e) const { return *reinterpret_cast<const f32x4*>(w + e); }
  __device__ __forceinline__ f32x4 shift(int, int e) const { return *reinterpret_cast<const f32x4*>(b + e); }
};
template <int NV>
struct LnParamsInRegs {
  static constexpr bool dpp_sums = true;  // 2 - 3 waves / SIMD walking rows one after the other: the reductions' latency is on the critical path
  const f32x4 (&w)[NV];
  const f32x4 (&b)[NV];
  __device__ __forceinline__ f32x4 scale(int i, int) const { return w[i]; }
  __device__ __forceinline__ f32x4 shift(int i, int) const { return b[i]; }
};

template <int NV, class Params>
__device__ __forceinline__ void ln_store_with(f32x4 (&v)[NV], int lane, int d, const Params prm, float eps, float* __restrict__ out,
                                              const StltDrop dr, uint32_t site, uint64_t row_idx0) {
#pragma clang fp contract(off)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float inv_d = 1.0f / (float)d;
  const float mean = (Params::dpp_sums ? wave_sum_dpp(s) : wave_sum(s)) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) {
      v[i] -= mean;
      q += fmaf(v[i].x, v[i].x, v[i].y * v[i].y) + fmaf(v[i].z, v[i].z, v[i].w * v[i].w);
    }
  }
  const float var = (Params::dpp_sums ? wave_sum_dpp(q) : wave_sum(q)) * inv_d;  // biased, as nn.LayerNorm
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int e = (i * 64 + lane) * 4;
    if (e < d) {
      const f32x4 ww = prm.scale(i, e), bb = prm.shift(i, e);
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = fmaf(v[i][c] * rstd, ww[c], bb[c]);
      if (dr.thr) o = stlt_drop4(dr, site, row_idx0 + e, o);  // train-mode dropout on the LayerNorm output
      *reinterpret_cast<f32x4*>(out + e) = o;
    }
  }
}

template <int NV>
__device__ __forceinline__ void ln_store(f32x4 (&v)[NV], int lane, int d, const float* __restrict__ w,
                                         const float* __restrict__ b, float eps, float* __restrict__ out,
                                         const StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0, uint64_t row_idx0 = 0) {
  ln_store_with<NV>(v, lane, d, LnParamsInMemory{w, b}, eps, out, dr, site, row_idx0);
}

template <int NV>
__device__ __forceinline__ void ln_store_regs(f32x4 (&v)[NV], int lane, int d, const f32x4 (&ww)[NV], const f32x4 (&bb)[NV], float eps,
                                              float* __restrict__ out, const StltDrop dr, uint32_t site, uint64_t row_idx0) {
  ln_store_with<NV>(v, lane, d, LnParamsInRegs<NV>{ww, bb}, eps, out, dr, site, row_idx0);
}

// one vector of a token's pre-LayerNorm embedding: table row + box projection (+ score projection) — models.py:29-39; written out as above
__device__ __forceinline__ f32x4 embed_value(f32x4 x, const f32x4 box, const f32x4 (&wr)[4], const f32x4 bb, bool with_score, float sc,
                                             const f32x4 ws, const f32x4 bs) {
#pragma clang fp contract(off)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float lin = (fmaf(box.x, wr[c].x, box.y * wr[c].y) + fmaf(box.z, wr[c].z, box.w * wr[c].w)) + bb[c];
    x[c] += lin;
    if (with_score) x[c] += fmaf(sc, ws[c], bs[c]);
  }
  return x;
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
      const f32x4 bb = *reinterpret_cast<const f32x4*>(box_b + e);
      f32x4 wr[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) wr[c] = *reinterpret_cast<const f32x4*>(box_w + (int64_t)(e + c) * 4);  // box_w is (d,4)
      f32x4 ws = f32x4{0.f, 0.f, 0.f, 0.f}, bs = ws;
      if (scores) {
        ws = *reinterpret_cast<const f32x4*>(score_w + e);  // (d,1)
        bs = *reinterpret_cast<const f32x4*>(score_b + e);
      }
      x = embed_value(x, box, wr, bb, scores != nullptr, sc, ws, bs);
      v[i] = x;
      if (pre_out) *reinterpret_cast<f32x4*>(pre_out + row * d + e) = x;  // training tape: pre-LayerNorm sum
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  ln_store<NV>(v, lane, d, ln_w, ln_b, eps, out + row * d, dr, STLT_SITE_EMBED, (uint64_t)row * d);
}

// K1 for many tokens: a wave embeds EMB_ROWS consecutive tokens with the box projection (d x 4), its bias, the score projection and the
// LayerNorm scale / shift in registers.  At one token per wave (embed_kernel) these 7 - 9 parameter vectors are re-read from L1 / L2 per
// output vector, and the kernel is bound by that traffic, not by its HBM write (round 6: 272 us for the 229 376 tokens of 1024 clips, bound
// 114 us).  Lanes 0 .. EMB_ROWS-1 fetch the wave's token indices, categories, boxes and scores in one step, so a row's dependent chain is
// only table row -> LayerNorm -> store, and the next table row is in flight while the current one is normalised.  Same operation order
// per element as embed_kernel: bit-identical rows.
constexpr int EMB_ROWS = 8;
__device__ __forceinline__ float lane_value(float v, int uniform_lane) {  // v_readlane_b32: the lane index is wave-uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), uniform_lane));
}
template <int NV>
__global__ __launch_bounds__(256) void embed_rows_kernel(const int64_t* __restrict__ categories, const float* __restrict__ boxes,
                                                         const float* __restrict__ scores, const float* __restrict__ cat_table,
                                                         int n_categories, const float* __restrict__ box_w,
                                                         const float* __restrict__ box_b, const float* __restrict__ score_w,
                                                         const float* __restrict__ score_b, const float* __restrict__ ln_w,
                                                         const float* __restrict__ ln_b, float eps, int64_t n_tokens, int d,
                                                         float* __restrict__ out, float* __restrict__ pre_out, StltDrop dr,
                                                         const int* __restrict__ src_index) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * ROWS_PER_BLOCK + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * EMB_ROWS;  // scalar
  if (row0 >= n_tokens) return;
  const int n_rows = (int)(n_tokens - row0 < EMB_ROWS ? n_tokens - row0 : EMB_ROWS);
  // per-lane token record (lanes past n_rows repeat the wave's last token: loads stay in bounds, values unused)
  const int64_t my_row = row0 + (lane < n_rows ? lane : n_rows - 1);
  const int64_t my_src = src_index ? src_index[my_row] : my_row;
  int64_t my_cat = categories[my_src];
  my_cat = my_cat < 0 ? 0 : (my_cat >= n_categories ? n_categories - 1 : my_cat);  // never read outside the table
  const f32x4 my_box = *reinterpret_cast<const f32x4*>(boxes + my_src * 4);
  const float my_sc = scores ? scores[my_src] : 0.f;
  const int my_cat32 = (int)my_cat;
  f32x4 wr[NV][4], bbv[NV], wsv[NV], bsv[NV], lw[NV], lb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = (i * 64 + lane) * 4;
    if (e < d) {
#pragma unroll
      for (int c = 0; c < 4; ++c) wr[i][c] = *reinterpret_cast<const f32x4*>(box_w + (int64_t)(e + c) * 4);  // box_w is (d,4)
      bbv[i] = *reinterpret_cast<const f32x4*>(box_b + e);
      lw[i] = *reinterpret_cast<const f32x4*>(ln_w + e);
      lb[i] = *reinterpret_cast<const f32x4*>(ln_b + e);
      if (scores) {
        wsv[i] = *reinterpret_cast<const f32x4*>(score_w + e);  // (d,1)
        bsv[i] = *reinterpret_cast<const f32x4*>(score_b + e);
      }
    }
  }
  f32x4 xn[NV];  // table row of the token about to be processed
  {
    const float* __restrict__ erow = cat_table + (int64_t)__builtin_amdgcn_readlane(my_cat32, 0) * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < d) xn[i] = *reinterpret_cast<const f32x4*>(erow + e);
    }
  }
  for (int r = 0; r < n_rows; ++r) {
    const int64_t row = row0 + r;
    f32x4 box;
    box.x = lane_value(my_box.x, r);
    box.y = lane_value(my_box.y, r);
    box.z = lane_value(my_box.z, r);
    box.w = lane_value(my_box.w, r);
    const float sc = lane_value(my_sc, r);
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = xn[i];
    if (r + 1 < n_rows) {
      const float* __restrict__ erow = cat_table + (int64_t)__builtin_amdgcn_readlane(my_cat32, r + 1) * d;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e = (i * 64 + lane) * 4;
        if (e < d) xn[i] = *reinterpret_cast<const f32x4*>(erow + e);
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < d) {
        const f32x4 x = embed_value(v[i], box, wr[i], bbv[i], scores != nullptr, sc, wsv[i], bsv[i]);
        v[i] = x;
        if (pre_out) *reinterpret_cast<f32x4*>(pre_out + row * d + e) = x;  // training tape: pre-LayerNorm sum
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    ln_store_regs<NV>(v, lane, d, lw, lb, eps, out + row * d, dr, STLT_SITE_EMBED, (uint64_t)row * d);
  }
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
  // STLT_EMBED_ROWS: token count from which a wave embeds EMB_ROWS tokens with the parameters in registers (0: never); below it the
  // launch is too small to fill the chip with 1/8 of the waves, and the one-token-per-wave kernel is as fast
  static const int64_t rows_from = [] { const char* e = getenv("STLT_EMBED_ROWS"); return e ? (int64_t)atoll(e) : (int64_t)32768; }();
  if (rows_from > 0 && n_tokens >= rows_from && d <= 1024) {
    const int64_t per_block = (int64_t)ROWS_PER_BLOCK * EMB_ROWS;
    dim3 grid_rows((unsigned)((n_tokens + per_block - 1) / per_block));
    switch (nv_for(d)) {
      case 1: hipLaunchKernelGGL((embed_rows_kernel<1>), grid_rows, dim3(256), 0, s, categories, boxes, scores, cat_table, (int)n_categories, box_w, box_b, score_w, score_b, ln_w, ln_b, eps, n_tokens, (int)d, out, pre_out, dr, src_index); break;
      case 2: hipLaunchKernelGGL((embed_rows_kernel<2>), grid_rows, dim3(256), 0, s, categories, boxes, scores, cat_table, (int)n_categories, box_w, box_b, score_w, score_b, ln_w, ln_b, eps, n_tokens, (int)d, out, pre_out, dr, src_index); break;
      case 3: hipLaunchKernelGGL((embed_rows_kernel<3>), grid_rows, dim3(256), 0, s, categories, boxes, scores, cat_table, (int)n_categories, box_w, box_b, score_w, score_b, ln_w, ln_b, eps, n_tokens, (int)d, out, pre_out, dr, src_index); break;
      default: hipLaunchKernelGGL((embed_rows_kernel<4>), grid_rows, dim3(256), 0, s, categories, boxes, scores, cat_table, (int)n_categories, box_w, box_b, score_w, score_b, ln_w, ln_b, eps, n_tokens, (int)d, out, pre_out, dr, src_index); break;
    }
    return stlt_check_launch("embed_rows_kernel");
  }
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
