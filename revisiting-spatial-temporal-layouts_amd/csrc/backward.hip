// Backward kernels of the STLT path (what autograd runs for the reference's train step, src/train.py:125-127):
// LayerNorm backward, bias-gradient column sums, GELU forward/backward, attention-core backward, and the
// backward of the embedding / frames-embedding / last-state gather.  All parameter-gradient reductions are
// two-stage (per-wave or per-block partials, then a fixed-order sum): bitwise reproducible, no float atomics.
// The large products (dX = dY·W, dW = dYᵀ·X) run on the MFMA kernel of gemm.hip.
#include <cstdlib>
#include "common.h"
#include "wave_dpp.h"

namespace {

constexpr int RW_WAVES = 4;  // waves per 256-thread block for the row-wise kernels
#ifndef STLT_LN_BWD_PIPE
#define STLT_LN_BWD_PIPE 1  // 0: a row's loads issued when its turn comes (A/B builds)
#endif

// ------------------------------------------------------------------ LayerNorm backward
// y = LN(s) * w + b with s = a (+ b2).  Given dy: ds = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*w;
// dw = sum_rows dy*xhat, db = sum_rows dy.  One wave per row (row in registers), persistent waves accumulate
// their dw/db partials in registers and write one partial row pair each.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                     const float* __restrict__ a, int64_t lda,
                                                     const float* __restrict__ b2, int64_t ldb,
                                                     const float* __restrict__ w, float eps, int64_t M, int d,
                                                     float* __restrict__ ds, int64_t ldds,
                                                     float* __restrict__ partials /* [blocks][3][d] */, StltDrop dr,
                                                     uint32_t site_b2, float* __restrict__ ds_drop, uint32_t site_dy,
                                                     const int* __restrict__ drop_rows) {
  __shared__ float red[3 * NV * 256];  // block-level sums of dw | db | column sums of the branch gradient
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * RW_WAVES + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * RW_WAVES;
  f32x4 dw_acc[NV], db_acc[NV], cs_acc[NV], wv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    dw_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    cs_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int e = (i * 64 + lane) * 4;
    wv[i] = e < d ? *reinterpret_cast<const f32x4*>(w + e) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv_d = 1.0f / (float)d;
  // The waves are persistent at 2 per SIMD (136+ registers, 512 blocks), each walking ~7 rows whose work is one dependent chain (three row
  // loads -> four wave-wide sums -> stores): the next row's loads are issued before the current row's arithmetic (STLT_LN_BWD_PIPE), and the
  // sums run on DPP / permlane swaps (wave_dpp.h: the shuffle butterfly's bits without its 24 LDS round trips per row).
  f32x4 na[NV], nb[NV], ng[NV];  // the next row as loaded: a, b2, dy
  auto load_row = [&](int64_t row) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < d) {
        na[i] = *reinterpret_cast<const f32x4*>(a + row * lda + e);
        if (b2) nb[i] = *reinterpret_cast<const f32x4*>(b2 + row * ldb + e);
        ng[i] = *reinterpret_cast<const f32x4*>(dy + row * lddy + e);
      }
    }
  };
  if (STLT_LN_BWD_PIPE && gw < M) load_row(gw);
  for (int64_t row = gw; row < M; row += n_waves) {
    const uint64_t drow = (dr.thr && drop_rows) ? (uint64_t)drop_rows[row] : (uint64_t)row;  // dropout masks follow the row's original position
    f32x4 x[NV], g[NV], bcur[NV];
    if (!STLT_LN_BWD_PIPE) load_row(row);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      x[i] = na[i];
      bcur[i] = nb[i];
      g[i] = ng[i];
    }
    if (STLT_LN_BWD_PIPE && row + n_waves < M) load_row(row + n_waves);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < d) {
        if (b2) {
          f32x4 bv = bcur[i];
          if (dr.thr && site_b2) bv = stlt_drop4(dr, site_b2, drow * d + e, bv);  // the forward added drop(b2)
          x[i] += bv;
        }
        if (dr.thr && site_dy) g[i] = stlt_drop4(dr, site_dy, drow * d + e, g[i]);  // dropout on the LN output
        sum += (x[i].x + x[i].y) + (x[i].z + x[i].w);
      } else {
        x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = wave_sum_dpp(sum) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < d) {
        x[i] -= mean;
        q += (x[i].x * x[i].x + x[i].y * x[i].y) + (x[i].z * x[i].z + x[i].w * x[i].w);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) * inv_d + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < d) {
        x[i] *= rstd;                 // xhat
        db_acc[i] += g[i];
        dw_acc[i] += g[i] * x[i];
        g[i] *= wv[i];                // g = dy * w
        sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        sgx += (g[i].x * x[i].x + g[i].y * x[i].y) + (g[i].z * x[i].z + g[i].w * x[i].w);
      }
    }
    const float mg = wave_sum_dpp(sg) * inv_d, mgx = wave_sum_dpp(sgx) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < d) {
        const f32x4 o = (g[i] - mg - x[i] * mgx) * rstd;
        *reinterpret_cast<f32x4*>(ds + row * ldds + e) = o;
        if (ds_drop) {
          const f32x4 od = stlt_drop4(dr, site_b2, drow * d + e, o);  // gradient wrt the un-dropped b2
          *reinterpret_cast<f32x4*>(ds_drop + row * ldds + e) = od;
          cs_acc[i] += od;
        } else {
          cs_acc[i] += o;
        }
      }
    }
  }
  if (partials) {  // the block's waves add their register partials in wave order (deterministic), one partial row set per block
    const int wave = threadIdx.x >> 6;
    for (int w = 0; w < RW_WAVES; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int e = (i * 64 + lane) * 4;
          if (e < d) {
            f32x4* r0 = reinterpret_cast<f32x4*>(red + e);
            f32x4* r1 = reinterpret_cast<f32x4*>(red + d + e);
            f32x4* r2 = reinterpret_cast<f32x4*>(red + 2 * d + e);
            if (w == 0) { *r0 = dw_acc[i]; *r1 = db_acc[i]; *r2 = cs_acc[i]; }
            else { *r0 += dw_acc[i]; *r1 += db_acc[i]; *r2 += cs_acc[i]; }
          }
        }
      }
      __syncthreads();
    }
    for (int e = threadIdx.x * 4; e < 3 * d; e += 256 * 4)
      *reinterpret_cast<f32x4*>(partials + (int64_t)blockIdx.x * 3 * d + e) = *reinterpret_cast<const f32x4*>(red + e);
  }
}

// The same for rows of 1025 .. 2048 channels (no shape of the path: every released model has d = 768; kept so that the training
// entry points take what the forward takes).  With such a row and its three accumulator rows in registers the kernel above needed
// 512 VGPRs, 659 spilled registers and 728 B of scratch per lane; here the per-wave partial rows live in dynamic LDS
// (4 waves x 3 x d floats) and a wave walks its row four times (mean, variance, the two gradient sums, the output), re-reading it
// from the cache instead of holding it.  Rows in row order per wave, waves in wave order per block: a fixed summation order.
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ a, int64_t lda,
                                                          const float* __restrict__ b2, int64_t ldb, const float* __restrict__ w, float eps, int64_t M,
                                                          int d, float* __restrict__ ds, int64_t ldds, float* __restrict__ partials, StltDrop dr,
                                                          uint32_t site_b2, float* __restrict__ ds_drop, uint32_t site_dy,
                                                          const int* __restrict__ drop_rows) {
  extern __shared__ __attribute__((aligned(16))) float lds_acc[];  // [wave][dw | db | colsum][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t gw = (int64_t)blockIdx.x * RW_WAVES + wave;
  const int64_t n_waves = (int64_t)gridDim.x * RW_WAVES;
  float* acc = lds_acc + wave * 3 * d;
#pragma unroll 1
  for (int e = lane * 4; e < 3 * d; e += 256) *reinterpret_cast<f32x4*>(acc + e) = f32x4{0.f, 0.f, 0.f, 0.f};
  const float inv_d = 1.0f / (float)d;
  for (int64_t row = gw; row < M; row += n_waves) {
    const uint64_t drow = (dr.thr && drop_rows) ? (uint64_t)drop_rows[row] : (uint64_t)row;
    auto load_s = [&](int e) {  // the LayerNorm input a + drop(b2)
      f32x4 v = *reinterpret_cast<const f32x4*>(a + row * lda + e);
      if (b2) {
        f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + row * ldb + e);
        if (dr.thr && site_b2) bv = stlt_drop4(dr, site_b2, drow * d + e, bv);
        v += bv;
      }
      return v;
    };
    auto load_g = [&](int e) {  // the output gradient behind the dropout on the LN output
      f32x4 v = *reinterpret_cast<const f32x4*>(dy + row * lddy + e);
      if (dr.thr && site_dy) v = stlt_drop4(dr, site_dy, drow * d + e, v);
      return v;
    };
    float sum = 0.f;
#pragma unroll 1
    for (int e = lane * 4; e < d; e += 256) { const f32x4 v = load_s(e); sum += (v.x + v.y) + (v.z + v.w); }
    const float mean = wave_sum(sum) * inv_d;
    float q = 0.f;
#pragma unroll 1
    for (int e = lane * 4; e < d; e += 256) { const f32x4 v = load_s(e) - mean; q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll 1
    for (int e = lane * 4; e < d; e += 256) {
      const f32x4 xh = (load_s(e) - mean) * rstd;
      f32x4 g = load_g(e);
      *reinterpret_cast<f32x4*>(acc + d + e) += g;
      *reinterpret_cast<f32x4*>(acc + e) += g * xh;
      g *= *reinterpret_cast<const f32x4*>(w + e);
      sg += (g.x + g.y) + (g.z + g.w);
      sgx += (g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w);
    }
    const float mg = wave_sum(sg) * inv_d, mgx = wave_sum(sgx) * inv_d;
#pragma unroll 1
    for (int e = lane * 4; e < d; e += 256) {
      const f32x4 xh = (load_s(e) - mean) * rstd;
      const f32x4 g = load_g(e) * *reinterpret_cast<const f32x4*>(w + e);
      const f32x4 o = (g - mg - xh * mgx) * rstd;
      *reinterpret_cast<f32x4*>(ds + row * ldds + e) = o;
      if (ds_drop) {
        const f32x4 od = stlt_drop4(dr, site_b2, drow * d + e, o);
        *reinterpret_cast<f32x4*>(ds_drop + row * ldds + e) = od;
        *reinterpret_cast<f32x4*>(acc + 2 * d + e) += od;
      } else {
        *reinterpret_cast<f32x4*>(acc + 2 * d + e) += o;
      }
    }
  }
  if (partials) {
    __syncthreads();
    for (int e = threadIdx.x * 4; e < 3 * d; e += 256 * 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(lds_acc + e);
#pragma unroll
      for (int w2 = 1; w2 < RW_WAVES; ++w2) t += *reinterpret_cast<const f32x4*>(lds_acc + w2 * 3 * d + e);
      *reinterpret_cast<f32x4*>(partials + (int64_t)blockIdx.x * 3 * d + e) = t;
    }
  }
}

// ------------------------------------------------------------------ column sums (bias gradients)
// partials[blockIdx.y][n] = sum over this block's row range of x[m][n]; 256 threads = 256 consecutive columns.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int N,
                                                             int64_t rows_per_block, float* __restrict__ partials) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  if (n >= N) return;
  // four independent loads in flight per thread, summed in row order (the result does not depend on the launch geometry of other kernels)
  float acc = 0.f;
  int64_t m = m0;
  for (; m + 4 <= m1; m += 4) {
    const float a0 = x[m * ld + n], a1 = x[(m + 1) * ld + n], a2 = x[(m + 2) * ld + n], a3 = x[(m + 3) * ld + n];
    acc += a0; acc += a1; acc += a2; acc += a3;
  }
  for (; m < m1; ++m) acc += x[m * ld + n];
  partials[(int64_t)blockIdx.y * N + n] = acc;
}

// ------------------------------------------------------------------ GELU (exact erf) forward / backward
// drop_rows (with ncols4 = row width / 4): rows picked out of a larger buffer keep the dropout masks of their
// original positions, i.e. element (row, col) uses index drop_rows[row] * width + col
__device__ __forceinline__ uint64_t drop_index(int64_t i4, const int* __restrict__ drop_rows, int64_t ncols4) {
  if (!drop_rows) return (uint64_t)i4 * 4;
  const int64_t row = i4 / ncols4;
  return ((uint64_t)drop_rows[row] * ncols4 + (uint64_t)(i4 - row * ncols4)) * 4;
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ u, float* __restrict__ h, int64_t n4,
                                                       StltDrop dr, uint32_t site, const int* __restrict__ drop_rows, int64_t ncols4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(u)[i];
    f32x4 o = {gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
    if (dr.thr) o = stlt_drop4(dr, site, drop_index(i, drop_rows, ncols4), o);
    reinterpret_cast<f32x4*>(h)[i] = o;
  }
}

__device__ __forceinline__ float gelu_grad(float x) {
  // d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x exp(-x^2/2) / sqrt(2 pi)
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * expf(-0.5f * x * x) * 0.39894228040143267794f;
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ u,
                                                       float* __restrict__ du, int64_t n4, StltDrop dr, uint32_t site,
                                                       const int* __restrict__ drop_rows, int64_t ncols4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 g = reinterpret_cast<const f32x4*>(dh)[i];
    if (dr.thr) g = stlt_drop4(dr, site, drop_index(i, drop_rows, ncols4), g);
    const f32x4 v = reinterpret_cast<const f32x4*>(u)[i];
    f32x4 o = {g.x * gelu_grad(v.x), g.y * gelu_grad(v.y), g.z * gelu_grad(v.z), g.w * gelu_grad(v.w)};
    reinterpret_cast<f32x4*>(du)[i] = o;
  }
}

// du = dh * gelu'(u) on an (M, N) matrix with the column sums of du (the bias gradient of the producing Linear)
// accumulated on the way: a block owns 1024 columns x a row range, thread = 4 columns.
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const float* __restrict__ dh, const float* __restrict__ u,
                                                              float* __restrict__ du, int64_t M, int N, int64_t rows_per_block,
                                                              float* __restrict__ partials, StltDrop dr, uint32_t site,
                                                              const int* __restrict__ drop_rows) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= N) return;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int64_t m = m0; m < m1; ++m) {
    const int64_t i = m * N + c;
    f32x4 g = *reinterpret_cast<const f32x4*>(dh + i);
    if (dr.thr) g = stlt_drop4(dr, site, drop_rows ? (uint64_t)drop_rows[m] * N + c : (uint64_t)i, g);
    const f32x4 v = *reinterpret_cast<const f32x4*>(u + i);
    const f32x4 o = {g.x * gelu_grad(v.x), g.y * gelu_grad(v.y), g.z * gelu_grad(v.z), g.w * gelu_grad(v.w)};
    *reinterpret_cast<f32x4*>(du + i) = o;
    acc += o;
  }
  *reinterpret_cast<f32x4*>(partials + (int64_t)blockIdx.y * N + c) = acc;
}

// ------------------------------------------------------------------ attention core backward
// One 256-thread block per (token group, head); a group = whole sequences totalling GL tokens (floor(32/L) sequences
// when L <= 32, one sequence when 32 < L <= 64).  Attention never crosses sequences, so scores / dS are stored per
// row as L columns (the row's own sequence) and every inner loop runs over L keys.  Operands live in dynamic LDS as
// fp32 (29-41 KB at L = 7 / 32: several blocks per CU); P is recomputed.
//   dV = P^T dO ; dP = dO V^T ; dS = P * (dP - rowsum(P*dP)) ; dQ = scale dS K ; dK = scale dS^T Q
constexpr int AB_MAXL = 64, AB_DH = 64, AB_LD = AB_DH + 1;

__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                       const uint8_t* __restrict__ kpm, int causal, int64_t n_tokens,
                                                       int L, int H, int GL, float scale, float* __restrict__ dqkv,
                                                       StltDrop dr, uint32_t site, int64_t n_groups,
                                                       float* __restrict__ cs_partials /* nullable: [chunks][3*H*64] */,
                                                       const int* __restrict__ grp_ptr, const int* __restrict__ seg_start,
                                                       const int* __restrict__ seg_end) {
  // Ragged mode (grp_ptr != null): group g = rows [grp_ptr[g], grp_ptr[g+1]) of a compacted buffer (whole segments,
  // at most GL rows); a row's sequence is [seg_start[row], seg_end[row]); L is the longest possible sequence.
  extern __shared__ float ab_smem[];
  const int PLD = L + 1;
  float* Qs = ab_smem;
  float* Ks = Qs + GL * AB_LD;
  float* Vs = Ks + GL * AB_LD;
  float* Gs = Vs + GL * AB_LD;
  float* Ps = Gs + GL * AB_LD;
  float* Ds = Ps + GL * PLD;
  int* keep = reinterpret_cast<int*>(Ds + GL * PLD);  // 1 = key token is real
  int* rs0 = keep + GL;                                // row -> first row of its sequence (group-local)
  int* rlen = rs0 + GL;                                // row -> length of its sequence (0 for rows past the group)
  const int tid = threadIdx.x;
  const int head = blockIdx.x % H;
  const int d = H * AB_DH;
  const int64_t ld = 3 * (int64_t)d;
  const int64_t chunk = blockIdx.x / H, n_chunks = gridDim.x / H;
  float cq = 0.f, ck = 0.f, cv = 0.f;  // column sums of dq/dk/dv over this thread's rows (channel tid&63): in-proj bias gradient
  for (int64_t g = chunk; g < n_groups; g += n_chunks) {
  const int64_t tok0 = grp_ptr ? (int64_t)grp_ptr[g] : g * GL;
  const int gv = grp_ptr ? grp_ptr[g + 1] - grp_ptr[g] : (int)((n_tokens - tok0) < GL ? (n_tokens - tok0) : GL);
  // load Q, K, V, dO (rows >= gv are zero)
  for (int idx = tid; idx < GL * AB_DH; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    float q = 0.f, k = 0.f, v = 0.f, go = 0.f;
    if (r < gv) {
      const float* row = qkv + (tok0 + r) * ld + head * AB_DH + c;
      q = row[0]; k = row[d]; v = row[2 * d];
      go = dctx[(tok0 + r) * (int64_t)d + head * AB_DH + c];
    }
    Qs[r * AB_LD + c] = q; Ks[r * AB_LD + c] = k; Vs[r * AB_LD + c] = v; Gs[r * AB_LD + c] = go;
  }
  if (tid < GL) {
    if (grp_ptr) {
      keep[tid] = tid < gv ? 1 : 0;
      rs0[tid] = tid < gv ? seg_start[tok0 + tid] - (int)tok0 : 0;
      rlen[tid] = tid < gv ? seg_end[tok0 + tid] - seg_start[tok0 + tid] : 0;
    } else {
      keep[tid] = (tid < gv && kpm[tok0 + tid] == 0) ? 1 : 0;
      rs0[tid] = (tid / L) * L;
      rlen[tid] = L;
    }
  }
  __syncthreads();
  // scores and dP for every (query i, key position jj of i's sequence)
  for (int p = tid; p < GL * L; p += 256) {
    const int i = p / L, jj = p - i * L;
    const int s0 = rs0[i], qp = i - s0;
    const bool in_seq = jj < rlen[i];
    const int j = in_seq ? s0 + jj : s0;
    const bool ok = in_seq && keep[j] && (!causal || jj <= qp);
    float s = 0.f, dp = 0.f;
#pragma unroll 8
    for (int c = 0; c < AB_DH; ++c) {
      s += Qs[i * AB_LD + c] * Ks[j * AB_LD + c];
      dp += Gs[i * AB_LD + c] * Vs[j * AB_LD + c];
    }
    if (dr.thr) {  // dPd -> dP: the forward multiplied P by the dropout mask before the P·V product
      const uint64_t idx = ((((uint64_t)(tok0 + i)) * H + head) << 8) | (uint64_t)(jj & 0xff);
      dp = stlt_keep(dr, site, idx) ? dp * dr.scale : 0.f;
    }
    Ps[i * PLD + jj] = ok ? s * scale : -1e30f;
    Ds[i * PLD + jj] = dp;
  }
  __syncthreads();
  // row softmax, D_i, dS (in place: Ps <- P, Ds <- dS)
  if (tid < GL) {
    const int i = tid;
    const int Li = rlen[i];
    float m = -1e30f;
    for (int j = 0; j < Li; ++j) m = fmaxf(m, Ps[i * PLD + j]);
    float l = 0.f;
    for (int j = 0; j < Li; ++j) {
      const float s = Ps[i * PLD + j];
      const float e = s > -1e29f ? expf(s - m) : 0.f;
      Ps[i * PLD + j] = e;
      l += e;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float dsum = 0.f;
    for (int j = 0; j < Li; ++j) {
      const float pj = Ps[i * PLD + j] * inv;
      Ps[i * PLD + j] = pj;
      dsum += pj * Ds[i * PLD + j];
    }
    for (int j = 0; j < Li; ++j) Ds[i * PLD + j] = Ps[i * PLD + j] * (Ds[i * PLD + j] - dsum);
  }
  __syncthreads();
  // dQ, dK, dV: thread owns (row r, channel c); consecutive threads -> consecutive channels (coalesced stores)
  for (int idx = tid; idx < GL * AB_DH; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    if (r >= gv) continue;
    const int s0 = rs0[r], rp = r - s0, Lr = rlen[r];
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int jj = 0; jj < Lr; ++jj) {
      const int j = s0 + jj;
      dq += Ds[r * PLD + jj] * Ks[j * AB_LD + c];   // sum_j dS[r][j] K[j][c]
      dk += Ds[j * PLD + rp] * Qs[j * AB_LD + c];   // sum_i dS[i][r] Q[i][c]
      float pd = Ps[j * PLD + rp];                  // P[query j][key r] (dropped like in the forward for dV)
      if (dr.thr) {
        const uint64_t idx = ((((uint64_t)(tok0 + j)) * H + head) << 8) | (uint64_t)(rp & 0xff);
        pd = stlt_keep(dr, site, idx) ? pd * dr.scale : 0.f;
      }
      dv += pd * Gs[j * AB_LD + c];                 // sum_i Pd[i][r] dO[i][c]
    }
    float* out = dqkv + (tok0 + r) * ld + head * AB_DH + c;
    out[0] = dq * scale;
    out[d] = dk * scale;
    out[2 * d] = dv;
    cq += dq * scale; ck += dk * scale; cv += dv;
  }
  __syncthreads();  // LDS is reloaded by the next group
  }
  if (cs_partials) {  // the 4 row classes (tid>>6) of a channel are added in order through LDS
    float* red = ab_smem;
    red[tid] = cq; red[256 + tid] = ck; red[512 + tid] = cv;
    __syncthreads();
    if (tid < 192) {
      const int which = tid >> 6, c = tid & 63;
      const float* r = red + which * 256 + c;
      cs_partials[chunk * 3 * d + which * d + head * AB_DH + c] = ((r[0] + r[64]) + r[128]) + r[192];
    }
  }
}

// Attention backward for sequences longer than AB_MAXL (up to AB_LONG_MAXL = 256, the position table's size): one
// block per (sequence, head) walks the queries in tiles of 32; per query tile, pass A streams the keys in tiles of 32
// through LDS for the scores and dP, the rows are normalised (whole rows of P / dS fit: 32 x (L+1) floats each), and
// pass B streams the keys again for dQ (registers) and for this tile's contribution to dK / dV, which accumulates in
// the output buffer (every (row, channel) element is owned by one thread, so the read-modify-write is private).
// Plain FMA code like attn_bwd_kernel: training with layouts of more than 64 frames is rare and this keeps it exact.
constexpr int AB_LONG_MAXL = 256;

__global__ __launch_bounds__(256) void attn_bwd_long_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                            const uint8_t* __restrict__ kpm, int causal, int L, int H, float scale,
                                                            float* __restrict__ dqkv, StltDrop dr, uint32_t site,
                                                            const int* __restrict__ grp_ptr) {
  extern __shared__ float ab_smem[];
  const int PLD = L + 1;
  float* Qs = ab_smem;                // 32 x 65: query tile
  float* Gs = Qs + 32 * AB_LD;        // dO tile
  float* Ks = Gs + 32 * AB_LD;        // key tile
  float* Vs = Ks + 32 * AB_LD;        // value tile
  float* Ps = Vs + 32 * AB_LD;        // 32 x (L+1): scores -> P
  float* Ds = Ps + 32 * PLD;          // dP -> dS
  const int tid = threadIdx.x;
  const int head = blockIdx.x % H;
  const int64_t g = blockIdx.x / H;
  const int64_t tok0 = grp_ptr ? (int64_t)grp_ptr[g] : g * L;
  const int len = grp_ptr ? grp_ptr[g + 1] - grp_ptr[g] : L;
  const int d = H * AB_DH;
  const int64_t ld = 3 * (int64_t)d;
  const int rc = tid >> 6, c = tid & 63;  // this thread's row class (rows r = rc, rc+4, ...) and channel
  const float* qb = qkv + tok0 * ld + head * AB_DH;
  float* ob = dqkv + tok0 * ld + head * AB_DH;
  for (int r = rc; r < len; r += 4) { ob[r * ld + d + c] = 0.f; ob[r * ld + 2 * d + c] = 0.f; }  // dK, dV accumulate below
  for (int q0 = 0; q0 < len; q0 += 32) {
    const int nq = len - q0 < 32 ? len - q0 : 32;
    for (int i = rc; i < 32; i += 4) {
      const bool in = i < nq;
      Qs[i * AB_LD + c] = in ? qb[(q0 + i) * ld + c] : 0.f;
      Gs[i * AB_LD + c] = in ? dctx[(tok0 + q0 + i) * (int64_t)d + head * AB_DH + c] : 0.f;
    }
    // ---- pass A: scores and dP against every key tile
    for (int k0 = 0; k0 < len; k0 += 32) {
      const int nkeys = len - k0 < 32 ? len - k0 : 32;
      __syncthreads();  // previous users of Ks / Vs (and the Q / dO tile writes above) are done
      for (int j = rc; j < 32; j += 4) {
        const bool in = j < nkeys;
        Ks[j * AB_LD + c] = in ? qb[(k0 + j) * ld + d + c] : 0.f;
        Vs[j * AB_LD + c] = in ? qb[(k0 + j) * ld + 2 * d + c] : 0.f;
      }
      __syncthreads();
      for (int p = tid; p < 32 * 32; p += 256) {
        const int i = p >> 5, j = p & 31;
        const int kj = k0 + j, qi = q0 + i;
        const bool real = j < nkeys && (grp_ptr ? true : kpm[tok0 + kj] == 0);
        const bool ok = i < nq && real && (!causal || kj <= qi);
        float sc = 0.f, dp = 0.f;
#pragma unroll 8
        for (int e = 0; e < AB_DH; ++e) {
          sc += Qs[i * AB_LD + e] * Ks[j * AB_LD + e];
          dp += Gs[i * AB_LD + e] * Vs[j * AB_LD + e];
        }
        if (dr.thr) {
          const uint64_t idx = ((((uint64_t)(tok0 + qi)) * H + head) << 8) | (uint64_t)(kj & 0xff);
          dp = stlt_keep(dr, site, idx) ? dp * dr.scale : 0.f;
        }
        if (j < nkeys) { Ps[i * PLD + kj] = ok ? sc * scale : -1e30f; Ds[i * PLD + kj] = dp; }
      }
    }
    __syncthreads();
    if (tid < 32) {  // row softmax, D_i, dS (in place: Ps <- P, Ds <- dS)
      const int i = tid;
      float m = -1e30f;
      for (int j = 0; j < len; ++j) m = fmaxf(m, Ps[i * PLD + j]);
      float l = 0.f;
      for (int j = 0; j < len; ++j) {
        const float sv = Ps[i * PLD + j];
        const float e = sv > -1e29f ? expf(sv - m) : 0.f;
        Ps[i * PLD + j] = e;
        l += e;
      }
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      float dsum = 0.f;
      for (int j = 0; j < len; ++j) {
        const float pj = Ps[i * PLD + j] * inv;
        Ps[i * PLD + j] = pj;
        dsum += pj * Ds[i * PLD + j];
      }
      for (int j = 0; j < len; ++j) Ds[i * PLD + j] = Ps[i * PLD + j] * (Ds[i * PLD + j] - dsum);
    }
    // ---- pass B: dQ of this query tile, and its share of dK / dV
    float dq[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) dq[t] = 0.f;
    for (int k0 = 0; k0 < len; k0 += 32) {
      const int nkeys = len - k0 < 32 ? len - k0 : 32;
      __syncthreads();  // the normalised rows are published; the previous tile's Ks readers are done
      for (int j = rc; j < 32; j += 4) Ks[j * AB_LD + c] = j < nkeys ? qb[(k0 + j) * ld + d + c] : 0.f;
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int i = rc + 4 * t;
        float acc = 0.f;
        for (int j = 0; j < nkeys; ++j) acc += Ds[i * PLD + k0 + j] * Ks[j * AB_LD + c];
        dq[t] += acc;
      }
      for (int j = rc; j < nkeys; j += 4) {
        const int kj = k0 + j;
        float dk = 0.f, dv = 0.f;
        for (int i = 0; i < nq; ++i) {
          dk += Ds[i * PLD + kj] * Qs[i * AB_LD + c];
          float pd = Ps[i * PLD + kj];
          if (dr.thr) {
            const uint64_t idx = ((((uint64_t)(tok0 + q0 + i)) * H + head) << 8) | (uint64_t)(kj & 0xff);
            pd = stlt_keep(dr, site, idx) ? pd * dr.scale : 0.f;
          }
          dv += pd * Gs[i * AB_LD + c];
        }
        ob[kj * ld + d + c] += dk * scale;  // same thread owns (kj, c) in every query tile
        ob[kj * ld + 2 * d + c] += dv;
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = rc + 4 * t;
      if (i < nq) ob[(q0 + i) * ld + c] = dq[t] * scale;
    }
    __syncthreads();  // the next query tile overwrites Qs / Gs / Ps / Ds
  }
}

// ------------------------------------------------------------------ K1 backward (parameter gradients)
// dx = gradient wrt the pre-LayerNorm embedding sum (tok, d).  Per block: a chunk of tokens; thread = channel.
// partial layout per block: [C category rows][4 box_w rows][box_b][score_w][score_b] x d
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ categories,
                                                        const float* __restrict__ boxes, const float* __restrict__ scores,
                                                        int C, int64_t n_tokens, int d, int64_t tok_per_block,
                                                        float* __restrict__ partials, const int* __restrict__ src_index) {
  // block = (token chunk, 256-channel slice); the per-category sums of the thread's channel live in LDS ([C][256]: a thread
  // only ever touches its own column, so plain read-modify-writes, no conflicts), the box / score sums in registers
  extern __shared__ float cat_acc[];
  const int64_t t0 = (int64_t)blockIdx.x * tok_per_block;
  const int64_t t1 = t0 + tok_per_block < n_tokens ? t0 + tok_per_block : n_tokens;
  float* out = partials + (int64_t)blockIdx.x * (C + 7) * d;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= d) return;
  for (int k = 0; k < C; ++k) cat_acc[k * 256 + threadIdx.x] = 0.f;
  float bw0 = 0.f, bw1 = 0.f, bw2 = 0.f, bw3 = 0.f, bb = 0.f, sw = 0.f;
  for (int64_t t = t0; t < t1; ++t) {
    const float g = dx[t * d + c];
    const int64_t src = src_index ? src_index[t] : t;  // ragged: gradient row -> token of the padded batch
    const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + src * 4);
    bw0 += g * bx.x; bw1 += g * bx.y; bw2 += g * bx.z; bw3 += g * bx.w;
    bb += g;
    if (scores) sw += g * scores[src];
    int64_t cat = categories[src];
    cat = cat < 0 ? 0 : (cat >= C ? C - 1 : cat);
    cat_acc[(int)cat * 256 + threadIdx.x] += g;
  }
  for (int k = 0; k < C; ++k) out[(int64_t)k * d + c] = cat_acc[k * 256 + threadIdx.x];
  out[(int64_t)(C + 0) * d + c] = bw0; out[(int64_t)(C + 1) * d + c] = bw1;
  out[(int64_t)(C + 2) * d + c] = bw2; out[(int64_t)(C + 3) * d + c] = bw3;
  out[(int64_t)(C + 4) * d + c] = bb;
  out[(int64_t)(C + 5) * d + c] = sw;
  out[(int64_t)(C + 6) * d + c] = bb;  // score_b gradient equals box_b's (both are plain sums)
}

// sum the block partials in block order and scatter into the parameter-gradient tensors (accumulating)
__global__ __launch_bounds__(256) void embed_bwd_finalize_kernel(const float* __restrict__ partials, int n_blocks, int C,
                                                                 int d, int has_scores, float* __restrict__ g_cat,
                                                                 float* __restrict__ g_box_w, float* __restrict__ g_box_b,
                                                                 float* __restrict__ g_score_w,
                                                                 float* __restrict__ g_score_b) {
  // block = (16 channels, one partial row kind); 16 lanes per channel each sum every 16th block partial, then a fixed-order
  // LDS pass adds the 16 lanes: deterministic, and 16x the parallelism of one thread per channel
  __shared__ float part[16][17];
  const int row = blockIdx.y;  // 0..C+6
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + col;
  float acc = 0.f;
  if (c < d)
    for (int b = sl; b < n_blocks; b += 16) acc += partials[((int64_t)b * (C + 7) + row) * d + c];
  part[sl][col] = acc;
  __syncthreads();
  if (sl != 0 || c >= d) return;
  acc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc += part[k][col];
  if (row < C) { if (g_cat && row != 0) g_cat[(int64_t)row * d + c] += acc; }   // padding_idx = 0: no gradient (models.py:22)
  else if (row < C + 4) { if (g_box_w) g_box_w[(int64_t)c * 4 + (row - C)] += acc; }  // box_w is (d,4)
  else if (row == C + 4) { if (g_box_b) g_box_b[c] += acc; }
  else if (row == C + 5) { if (has_scores && g_score_w) g_score_w[c] += acc; }    // (d,1)
  else { if (has_scores && g_score_b) g_score_b[c] += acc; }
}

// ------------------------------------------------------------------ K7 backward
// ds (B*T, d) = gradient wrt the pre-LayerNorm frames sum.  d_pos[t] = sum_b ds[b,t]; d_type[ft] = sum ds (row 0 =
// padding_idx gets none, models.py:91); the CLS rows of the spatial gradient receive ds, the other rows zero.
__global__ __launch_bounds__(256) void frames_bwd_scatter_kernel(const float* __restrict__ ds, int64_t BT, int N, int d,
                                                                 float* __restrict__ dx_spatial) {
  // one block per frame: row 0 <- ds, rows 1..N-1 <- 0
  const int64_t f = blockIdx.x;
  for (int idx = threadIdx.x * 4; idx < N * d; idx += 1024) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (idx < d) v = *reinterpret_cast<const f32x4*>(ds + f * d + idx);
    *reinterpret_cast<f32x4*>(dx_spatial + f * (int64_t)N * d + idx) = v;
  }
}

__global__ __launch_bounds__(256) void frames_bwd_params_kernel(const float* __restrict__ ds,
                                                                const int64_t* __restrict__ frame_types, int64_t B, int T,
                                                                int d, float* __restrict__ partials /* [chunks][T+5][d] */,
                                                                const int* __restrict__ row_of /* ragged: frame b*T+t -> row of ds, -1 = padded */) {
  // grid = (column blocks, T position rows followed by 5 type rows, clip chunks); a block sums its chunk of clips in
  // clip order, the chunk partials are added in chunk order afterwards: fixed summation order
  const int row = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const int64_t per = (B + gridDim.z - 1) / gridDim.z;
  const int64_t b0 = (int64_t)blockIdx.z * per, b1 = b0 + per < B ? b0 + per : B;
  float acc = 0.f;
  if (row < T) {
    for (int64_t b = b0; b < b1; ++b) {
      const int64_t r = row_of ? row_of[b * T + row] : b * T + row;
      if (r >= 0) acc += ds[r * d + c];
    }
  } else if (row - T != 0) {  // frame type 0 is the padding index: no gradient (models.py:91)
    const int ft = row - T;
    for (int64_t i = b0 * T; i < b1 * T; ++i) {
      int64_t v = frame_types[i];
      v = v < 0 ? 0 : (v > 4 ? 4 : v);
      const int64_t r = row_of ? row_of[i] : i;
      if (v == ft && r >= 0) acc += ds[r * d + c];
    }
  }
  partials[((int64_t)blockIdx.z * (T + 5) + row) * d + c] = acc;
}

// ------------------------------------------------------------------ K8a backward: scatter the head's input gradient
__global__ __launch_bounds__(256) void scatter_last_kernel(const float* __restrict__ dh, const int64_t* __restrict__ lengths,
                                                           int T, int d, float* __restrict__ dout /* (B,T,d), pre-zeroed */) {
  const int64_t b = blockIdx.x;
  int64_t t = lengths[b] - 1;
  t = t < 0 ? t + T : t;
  t = t < 0 ? 0 : (t >= T ? T - 1 : t);
  for (int e = threadIdx.x * 4; e < d; e += 1024)
    *reinterpret_cast<f32x4*>(dout + (b * T + t) * (int64_t)d + e) = *reinterpret_cast<const f32x4*>(dh + b * d + e);
}

// ------------------------------------------------------------------ small dense products (prediction head: M = B rows, N = 174 ...)
// c[m][n] (+)= sum_k a[m*sam + k*sak] * b[k*sbk + n*sbn]; one thread per output element; any layout via strides.
__global__ __launch_bounds__(256) void small_gemm_kernel(const float* __restrict__ a, int64_t sam, int64_t sak,
                                                         const float* __restrict__ b, int64_t sbk, int64_t sbn,
                                                         float* __restrict__ c, int64_t ldc, int M, int N, int K,
                                                         int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
  // eight loads of each operand in flight per step (a thread's k-loop is a chain of L2 round trips: 174 dependent steps were 43 us for the
  // 64 x 768 input gradient of a 174-class head); the partial sums are added in a fixed order
  const float* ap = a + m * sam;
  const float* bp = b + n * sbn;
  float acc = 0.f;
  int k = 0;
  for (; k + 8 <= K; k += 8) {
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { av[j] = ap[(k + j) * sak]; bv[j] = bp[(k + j) * sbk]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(av[j], bv[j], acc);
  }
  for (; k < K; ++k) acc = fmaf(ap[k * sak], bp[k * sbk], acc);
  float* o = c + (int64_t)m * ldc + n;
  *o = accumulate ? *o + acc : acc;
}

inline int nv_for(int64_t d) { return (int)((d + 255) / 256); }

}  // namespace

#define DISPATCH_NV(nv, ...)                                   \
  switch (nv) {                                                \
    case 1: { constexpr int NV = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int NV = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int NV = 3; __VA_ARGS__; } break;      \
    case 4: { constexpr int NV = 4; __VA_ARGS__; } break;      \
    default: { constexpr int NV = 8; __VA_ARGS__; } break;     \
  }

#define DISPATCH_NV4(nv, ...)                                  \
  switch (nv) {                                                \
    case 1: { constexpr int NV = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int NV = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int NV = 3; __VA_ARGS__; } break;      \
    default: { constexpr int NV = 4; __VA_ARGS__; } break;     \
  }

// ds = dLN(dy; s = a (+ b2)); parameter gradients ACCUMULATE into g_w / g_b (either may be null).
// scratch: >= ln_bwd_scratch_floats(d) floats.
int64_t ln_bwd_scratch_floats(int64_t d) { return 512 * 3 * d; }

int launch_ln_bwd(const float* dy, int64_t lddy, const float* a, int64_t lda, const float* b2, int64_t ldb, const float* w,
                  float eps, int64_t M, int64_t d, float* ds, int64_t ldds, float* g_w, float* g_b, float* scratch,
                  hipStream_t s, StltDrop dr, uint32_t site_b2, float* ds_drop, uint32_t site_dy, float* g_colsum,
                  const int* drop_rows) {
  if (!dy || !a || !w || !ds || !scratch) return stlt_set_error(STLT_EINVAL, "ln_bwd: null pointer");
  if (d <= 0 || d % 4 || d > 2048) return stlt_set_error(STLT_EINVAL, "ln_bwd: bad d=%lld", (long long)d);
  if (M == 0) return 0;
  // ~4 rows per persistent wave until the cap binds; few rows (the temporal tower at 64 clips: 2048) get a wave per row — at 4 rows per wave
  // the launch is 128 workgroups for 256 CUs and a wave's four dependent row passes (22 us for 31 MB; STLT_LN_BWD_ROWS_PER_WAVE=4 restores it)
  static const int rpw_env = [] { const char* e = getenv("STLT_LN_BWD_ROWS_PER_WAVE"); return e ? atoi(e) : 0; }();
  const int64_t rpw = rpw_env > 0 ? rpw_env : (M <= 512 * RW_WAVES ? 1 : 4);
  int64_t blocks = (M + rpw * RW_WAVES - 1) / (rpw * RW_WAVES);
  if (blocks > 512) blocks = 512;  // one partial row set per block (scratch is sized for that)
  StltProfScope ps(STLT_K_LN_BWD, s);
  stlt_prof_note("ln_bwd rows=%lld d=%lld blocks=%lld", (long long)M, (long long)d, (long long)blocks);
  stlt_prof_add_bytes((double)M * 4.0 * d * 4.0);  // dy, the two summands (or the sum), ds
  if (nv_for(d) > 4) {  // d > 1024: accumulators in LDS (no shape of the path has such rows; kept working without scratch memory)
    const size_t lds = (size_t)RW_WAVES * 3 * d * sizeof(float);  // <= 96 KB
    static StltPerDeviceOnce lds_once;
    if (!lds_once.flag()) {
      if (hipError_t e = hipFuncSetAttribute((const void*)ln_bwd_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RW_WAVES * 3 * 2048 * (int)sizeof(float)); e != hipSuccess)
        return stlt_set_error((int)e, "ln_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      lds_once.flag() = true;
    }
    hipLaunchKernelGGL(ln_bwd_wide_kernel, dim3((unsigned)blocks), dim3(256), lds, s, dy, lddy, a, lda, b2, ldb, w, eps, M, (int)d, ds, ldds, scratch, dr,
                       site_b2, (dr.thr && site_b2) ? ds_drop : (float*)nullptr, site_dy, drop_rows);
  } else {
    DISPATCH_NV4(nv_for(d), hipLaunchKernelGGL((ln_bwd_kernel<NV>), dim3((unsigned)blocks), dim3(256), 0, s, dy, lddy, a, lda, b2,
                                               ldb, w, eps, M, (int)d, ds, ldds, scratch, dr, site_b2,
                                               (dr.thr && site_b2) ? ds_drop : (float*)nullptr, site_dy, drop_rows));
  }
  if (int e = stlt_check_launch("ln_bwd_kernel")) return e;
  // partial rows are interleaved [block][dw|db|colsum][d]: one strided reduction for the three destinations
  return launch_reduce_slabs3(scratch, 3 * d, (int)blocks, g_w, g_b, g_colsum, d, 1, s);
}

// g[n] += sum_m x[m][n]   (scratch >= 64*N floats)
int launch_colsum_acc(const float* x, int64_t ld, int64_t M, int64_t N, float* g, float* scratch, hipStream_t s) {
  StltProfScope ps(STLT_K_MISC, s);
  if (!x || !g || !scratch) return stlt_set_error(STLT_EINVAL, "colsum: null pointer");
  if (M == 0 || N == 0) return 0;
  // row ranges of at least 16 rows, at most 64 of them (scratch): a 2112 x 768 input used to run as 27 workgroups walking 235 rows each
  // (47 us, 27 launches per CACNF step); now 192 workgroups of 33 rows
  int parts = (int)((M + 15) / 16);
  if (parts > 64) parts = 64;
  const int64_t rows = (M + parts - 1) / parts;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)parts), dim3(256), 0, s, x, ld, M, (int)N,
                     rows, scratch);
  if (int e = stlt_check_launch("colsum_partial_kernel")) return e;
  return launch_reduce_slabs(scratch, N, parts, g, N, 1, s);
}

int launch_gelu_fwd(const float* u, float* h, int64_t n, hipStream_t s, StltDrop dr, uint32_t site, const int* drop_rows, int64_t ncols) {
  StltProfScope ps(STLT_K_GELU, s);
  stlt_prof_note("gelu_fwd n=%lld", (long long)n);
  stlt_prof_add_bytes(8.0 * (double)n);
  if (n % 4) return stlt_set_error(STLT_EINVAL, "gelu: element count must be a multiple of 4");
  if (n == 0) return 0;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (drop_rows && (ncols <= 0 || ncols % 4)) return stlt_set_error(STLT_EINVAL, "gelu: row width must be a positive multiple of 4");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, u, h, n / 4, dr, site, drop_rows, ncols / 4);
  return stlt_check_launch("gelu_fwd_kernel");
}

int launch_gelu_bwd_colsum(const float* dh, const float* u, float* du, int64_t M, int64_t N, float* g_colsum, float* scratch,
                           hipStream_t s, StltDrop dr, uint32_t site, const int* drop_rows) {
  StltProfScope ps(STLT_K_GELU, s);
  stlt_prof_note("gelu_bwd+colsum rows=%lld cols=%lld", (long long)M, (long long)N);
  stlt_prof_add_bytes(12.0 * (double)M * (double)N);
  if (!dh || !u || !du || !g_colsum || !scratch) return stlt_set_error(STLT_EINVAL, "gelu_bwd: null pointer");
  if (N % 4 || N > 0x7fffff00LL) return stlt_set_error(STLT_EINVAL, "gelu: column count must be a multiple of 4");
  if (M == 0 || N == 0) return 0;
  int64_t rows = 32;
  if ((M + rows - 1) / rows > 512) rows = (M + 511) / 512;  // at most 512 partial rows (scratch >= 512*N floats)
  const int64_t parts = (M + rows - 1) / rows;
  hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3((unsigned)((N + 1023) / 1024), (unsigned)parts), dim3(256), 0, s, dh, u, du, M, (int)N,
                     rows, scratch, dr, site, drop_rows);
  if (int e = stlt_check_launch("gelu_bwd_colsum_kernel")) return e;
  return launch_reduce_slabs(scratch, N, (int)parts, g_colsum, N, 1, s);
}

int launch_gelu_bwd(const float* dh, const float* u, float* du, int64_t n, hipStream_t s, StltDrop dr, uint32_t site, const int* drop_rows,
                    int64_t ncols) {
  StltProfScope ps(STLT_K_GELU, s);
  if (n % 4) return stlt_set_error(STLT_EINVAL, "gelu: element count must be a multiple of 4");
  if (n == 0) return 0;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (drop_rows && (ncols <= 0 || ncols % 4)) return stlt_set_error(STLT_EINVAL, "gelu: row width must be a positive multiple of 4");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dh, u, du, n / 4, dr, site, drop_rows, ncols / 4);
  return stlt_check_launch("gelu_bwd_kernel");
}

int launch_attn_bwd(const float* qkv, const float* dctx, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H,
                    int64_t dh, float* dqkv, hipStream_t s, StltDrop dr, uint32_t site, float* g_colsum, float* scratch,
                    const AttnBwdRagged* rg) {
  StltProfScope ps(STLT_K_ATTN_BWD, s);
  stlt_prof_note("attn_bwd S=%lld L=%lld H=%lld causal=%d%s", (long long)S, (long long)L, (long long)H, causal, rg ? " ragged" : "");
  stlt_prof_add_bytes(4.0 * (double)(S * L) * (double)(H * dh) * 7.0 + (double)(S * L));  // read qkv + dctx, write dqkv
  stlt_prof_note_flops(10.0 * (double)S * (double)H * (double)L * (double)L * (double)dh);   // S, dP, dV, dQ, dK
  if (!qkv || !dctx || (!kpm && !rg) || !dqkv) return stlt_set_error(STLT_EINVAL, "attn_bwd: null pointer");
  if (L <= 0 || L > AB_LONG_MAXL)
    return stlt_set_error(STLT_EINVAL, "attention backward supports sequences of at most %d tokens (got L=%lld)", AB_LONG_MAXL, (long long)L);
  if (S == 0) return 0;
  if (dh != AB_DH) {  // any other head dim: attn_any.hip, then the in-projection bias gradient as column sums of dqkv
    const int64_t d = H * dh;
    if (rg && rg->n_groups == 0) return 0;
    if (int e = launch_attn_any_bwd(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, dctx, kpm, rg ? rg->grp_ptr : nullptr, rg ? rg->seg_start : nullptr,
                                    rg ? rg->seg_end : nullptr, rg ? rg->max_rows : 0, causal, rg ? rg->n_groups : S, L, L, H, dh, dqkv, 3 * d,
                                    dqkv + d, dqkv + 2 * d, 3 * d, s, dr, site))
      return e;
    if (!g_colsum) return 0;
    if (!scratch) return stlt_set_error(STLT_EINVAL, "attn_bwd: column sums need scratch");
    return launch_colsum_acc(dqkv, 3 * d, rg ? rg->n_rows : S * L, 3 * d, g_colsum, scratch, s);
  }
  if (L > AB_MAXL) {  // long sequences: one block per (sequence, head), keys streamed in tiles
    const int64_t n_seq = rg ? rg->n_groups : S;
    if (rg && rg->max_rows > AB_LONG_MAXL) return stlt_set_error(STLT_EINVAL, "attn_bwd: group of %d rows unsupported", rg->max_rows);
    if (n_seq == 0) return 0;
    if (n_seq * H > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "attn_bwd: too many sequences");
    const size_t lds_long = ((size_t)4 * 32 * AB_LD + (size_t)2 * 32 * (L + 1)) * sizeof(float);
    static StltPerDeviceOnce long_once;  // a function attribute is set per device
    bool& long_opt_in = long_once.flag();
    if (!long_opt_in) {
      if (hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); e != hipSuccess)
        return stlt_set_error((int)e, "attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      long_opt_in = true;
    }
    hipLaunchKernelGGL(attn_bwd_long_kernel, dim3((unsigned)(n_seq * H)), dim3(256), lds_long, s, qkv, dctx, kpm, causal, (int)L, (int)H,
                       1.0f / sqrtf((float)dh), dqkv, dr, site, rg ? rg->grp_ptr : (const int*)nullptr);
    if (int e = stlt_check_launch("attn_bwd_long_kernel")) return e;
    if (g_colsum) {  // in-proj bias gradient: column sums of dqkv in a pass of their own (scratch >= 64 * 3 * H * dh floats)
      if (!scratch) return stlt_set_error(STLT_EINVAL, "attn_bwd: column sums need scratch");
      return launch_colsum_acc(dqkv, 3 * H * dh, rg ? rg->n_rows : S * L, 3 * H * dh, g_colsum, scratch, s);
    }
    return 0;
  }
  if (!rg && dh == 64) {  // padded layout, sequences of at most 64 tokens: the MFMA kernel (attn_bwd16.hip)
    if (g_colsum && !scratch) return stlt_set_error(STLT_EINVAL, "attn_bwd: column sums need scratch");
    bool taken = false;
    int slabs = 0;
    const int rc = launch_attn_bwd16(qkv, dctx, kpm, causal, S, L, H, dqkv, dr, site, scratch, g_colsum != nullptr, &slabs, s, &taken);
    if (rc != 0) return rc;
    if (taken) return g_colsum ? launch_reduce_slabs(scratch, 3 * H * dh, slabs, g_colsum, 3 * H * dh, 1, s) : 0;
  }
  if (rg && dh == 64) {  // ragged layout, groups of at most 64 rows: the same kernel with segment masks
    if (g_colsum && !scratch) return stlt_set_error(STLT_EINVAL, "attn_bwd: column sums need scratch");
    bool taken = false;
    int slabs = 0;
    const int rc = launch_attn_bwd16_ragged(qkv, dctx, *rg, causal, H, dqkv, dr, site, scratch, g_colsum != nullptr, &slabs, s, &taken);
    if (rc != 0) return rc;
    if (taken) return g_colsum ? launch_reduce_slabs(scratch, 3 * H * dh, slabs, g_colsum, 3 * H * dh, 1, s) : 0;
  }
  const int P = L <= 32 ? (int)(32 / L) : 1;
  const int GL = rg ? rg->max_rows : P * (int)L;
  const int64_t groups = rg ? rg->n_groups : (S + P - 1) / P;
  if (GL < 1 || GL > AB_MAXL) return stlt_set_error(STLT_EINVAL, "attn_bwd: group of %d rows unsupported", GL);
  if (groups == 0) return 0;
  if (g_colsum && !scratch) return stlt_set_error(STLT_EINVAL, "attn_bwd: column sums need scratch");
  int64_t chunks = groups < 256 ? groups : 256;  // persistent blocks per head (scratch >= 256 * 3 * H * dh floats)
  const size_t lds = ((size_t)4 * GL * AB_LD + (size_t)2 * GL * (L + 1) + 3 * GL) * sizeof(float);
  static StltPerDeviceOnce lds_once;  // > 64 KB of dynamic LDS (one 64-token sequence: 100 KB) needs the attribute, per device
  bool& lds_opt_in = lds_once.flag();
  if (!lds_opt_in) {
    if (hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); e != hipSuccess)
      return stlt_set_error((int)e, "attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    lds_opt_in = true;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(chunks * H)), dim3(256), lds, s, qkv, dctx, kpm, causal, rg ? rg->n_rows : S * L, (int)L,
                     (int)H, GL, 1.0f / sqrtf((float)dh), dqkv, dr, site, groups, g_colsum ? scratch : (float*)nullptr,
                     rg ? rg->grp_ptr : (const int*)nullptr, rg ? rg->seg_start : (const int*)nullptr,
                     rg ? rg->seg_end : (const int*)nullptr);
  if (int e = stlt_check_launch("attn_bwd_kernel")) return e;
  if (g_colsum) return launch_reduce_slabs(scratch, 3 * H * dh, (int)chunks, g_colsum, 3 * H * dh, 1, s);
  return 0;
}

int64_t embed_bwd_scratch_floats(int64_t n_tokens, int64_t C, int64_t d) {
  const int64_t blocks = (n_tokens + 31) / 32 > 512 ? 512 : (n_tokens + 31) / 32;  // as launch_embed_bwd cuts the tokens
  return (blocks > 0 ? blocks : 1) * (C + 7) * d;
}

int launch_embed_bwd(const float* dx, const int64_t* categories, const float* boxes, const float* scores, int64_t C,
                     int64_t n_tokens, int64_t d, float* g_cat, float* g_box_w, float* g_box_b, float* g_score_w,
                     float* g_score_b, float* scratch, hipStream_t s, const int* src_index) {
  StltProfScope ps(STLT_K_EMBED_BWD, s);
  if (!dx || !categories || !boxes || !scratch) return stlt_set_error(STLT_EINVAL, "embed_bwd: null pointer");
  if (n_tokens < 0 || n_tokens > 0x7fffffff || C <= 0 || d <= 0 || d % 4 != 0) return stlt_set_error(STLT_EINVAL, "embed_bwd: bad shape (tokens %lld, categories %lld, d %lld)", (long long)n_tokens, (long long)C, (long long)d);
  if (n_tokens == 0) return 0;
  int64_t blocks = (n_tokens + 31) / 32;  // short token chunks x channel slices: enough blocks for every CU (round 3: 112 -> 1344 at 64 clips)
  if (blocks > 512) blocks = 512;
  const int64_t tpb = (n_tokens + blocks - 1) / blocks;
  const size_t lds = (size_t)C * 256 * sizeof(float);
  if (lds > 128 * 1024) return stlt_set_error(STLT_EINVAL, "embed_bwd: at most 128 categories (got %lld)", (long long)C);
  if (lds > 48 * 1024) {
    static StltPerDeviceOnce lds_once;
    if (!lds_once.flag()) {
      if (hipError_t e = hipFuncSetAttribute((const void*)embed_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); e != hipSuccess)
        return stlt_set_error((int)e, "embed_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      lds_once.flag() = true;
    }
  }
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)blocks, (unsigned)((d + 255) / 256)), dim3(256), lds, s, dx, categories, boxes, scores, (int)C,
                     n_tokens, (int)d, tpb, scratch, src_index);
  if (int e = stlt_check_launch("embed_bwd_kernel")) return e;
  hipLaunchKernelGGL(embed_bwd_finalize_kernel, dim3((unsigned)((d + 15) / 16), (unsigned)(C + 7)), dim3(256), 0, s, scratch,
                     (int)blocks, (int)C, (int)d, scores ? 1 : 0, g_cat, g_box_w, g_box_b, g_score_w, g_score_b);
  return stlt_check_launch("embed_bwd_finalize_kernel");
}

int launch_frames_bwd(const float* ds, const int64_t* frame_types, int64_t B, int64_t T, int64_t N, int64_t d,
                      float* dx_spatial, float* g_pos, float* g_type, float* scratch, hipStream_t s, const int* row_of) {
  StltProfScope ps(STLT_K_EMBED_BWD, s);
  if (!ds || !frame_types || !scratch) return stlt_set_error(STLT_EINVAL, "frames_bwd: null pointer");
  if (B * T == 0) return 0;
  if (dx_spatial) {  // null: the caller routes the CLS-row gradient itself (ragged layout / CLS-rows-only last layer)
    hipLaunchKernelGGL(frames_bwd_scatter_kernel, dim3((unsigned)(B * T)), dim3(256), 0, s, ds, B * T, (int)N, (int)d, dx_spatial);
    if (int e = stlt_check_launch("frames_bwd_scatter_kernel")) return e;
  }
  const int chunks = B < 16 ? (int)B : 16;  // scratch >= 16 * (T+5) * d floats
  hipLaunchKernelGGL(frames_bwd_params_kernel, dim3((unsigned)((d + 255) / 256), (unsigned)(T + 5), (unsigned)chunks), dim3(256), 0, s, ds,
                     frame_types, B, (int)T, (int)d, scratch, row_of);
  if (int e = stlt_check_launch("frames_bwd_params_kernel")) return e;
  if (g_pos) { if (int e = launch_reduce_slabs(scratch, (T + 5) * d, chunks, g_pos, T * d, 1, s)) return e; }
  if (g_type) { if (int e = launch_reduce_slabs(scratch + T * d, (T + 5) * d, chunks, g_type, 5 * d, 1, s)) return e; }
  return 0;
}

int launch_scatter_last(const float* dh, const int64_t* lengths, int64_t B, int64_t T, int64_t d, float* dout, hipStream_t s) {
  StltProfScope ps(STLT_K_MISC, s);
  if (!dh || !lengths || !dout) return stlt_set_error(STLT_EINVAL, "scatter_last: null pointer");
  if (B == 0) return 0;
  if (hipError_t e = hipMemsetAsync(dout, 0, (size_t)B * T * d * sizeof(float), s); e != hipSuccess)
    return stlt_set_error((int)e, "scatter_last: memset: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(scatter_last_kernel, dim3((unsigned)B), dim3(256), 0, s, dh, lengths, (int)T, (int)d, dout);
  return stlt_check_launch("scatter_last_kernel");
}

int launch_small_gemm(const float* a, int64_t sam, int64_t sak, const float* b, int64_t sbk, int64_t sbn, float* c,
                      int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, hipStream_t s) {
  StltProfScope ps(STLT_K_MISC, s);
  if (!a || !b || !c) return stlt_set_error(STLT_EINVAL, "small_gemm: null pointer");
  if (M * N == 0) return 0;
  hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, a, sam, sak, b, sbk, sbn, c, ldc,
                     (int)M, (int)N, (int)K, accumulate);
  return stlt_check_launch("small_gemm_kernel");
}
