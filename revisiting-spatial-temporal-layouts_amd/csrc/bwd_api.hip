// Per-kernel backward entry points of the C-ABI (the *_bwd counterparts of the K-row forwards): what an op-level
// autograd needs — the fusion models' training step is composed from these in Python (modelling/fusion.py), while the
// STLT training step keeps its single fixed reverse sweep (train.hip).
#include "common.h"

namespace {

constexpr int XB_MAXL = 64, XB_DH = 64, XB_LD = XB_DH + 1;

// Backward of the attention core for separate query / key-value token spaces (cross-attention; self-attention on a
// packed buffer is q = qkv, k = qkv + d, v = qkv + 2d).  One block per (sequence, head), everything in LDS as fp32,
// P recomputed.  dV = Pᵀ dO ; dP = dO Vᵀ ; dS = P ∘ (dP − rowsum(P ∘ dP)) ; dQ = scale dS K ; dK = scale dSᵀ Q.
__global__ __launch_bounds__(256) void attn_bwd_general_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                               const float* __restrict__ v, int64_t ldkv,
                                                               const float* __restrict__ dctx, const uint8_t* __restrict__ kpm,
                                                               int causal, int Lq, int Lk, int H, float scale,
                                                               float* __restrict__ dq, int64_t lddq, float* __restrict__ dk,
                                                               float* __restrict__ dv, int64_t lddkv, StltDrop dr, uint32_t site) {
  extern __shared__ float xb_smem[];
  const int PLD = Lk + 1;
  float* Qs = xb_smem;
  float* Gs = Qs + Lq * XB_LD;
  float* Ks = Gs + Lq * XB_LD;
  float* Vs = Ks + Lk * XB_LD;
  float* Ps = Vs + Lk * XB_LD;
  float* Ds = Ps + Lq * PLD;
  const int tid = threadIdx.x;
  const int head = blockIdx.x % H;
  const int64_t sq = blockIdx.x / H;
  const int d = H * XB_DH;
  const int64_t q0 = sq * Lq, k0 = sq * Lk;
  for (int idx = tid; idx < Lq * XB_DH; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    Qs[r * XB_LD + c] = q[(q0 + r) * ldq + head * XB_DH + c];
    Gs[r * XB_LD + c] = dctx[(q0 + r) * (int64_t)d + head * XB_DH + c];
  }
  for (int idx = tid; idx < Lk * XB_DH; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    Ks[r * XB_LD + c] = k[(k0 + r) * ldkv + head * XB_DH + c];
    Vs[r * XB_LD + c] = v[(k0 + r) * ldkv + head * XB_DH + c];
  }
  __syncthreads();
  for (int p = tid; p < Lq * Lk; p += 256) {
    const int i = p / Lk, j = p - i * Lk;
    const bool ok = (!kpm || kpm[k0 + j] == 0) && (!causal || j <= i);
    float s = 0.f, dp = 0.f;
#pragma unroll 8
    for (int c = 0; c < XB_DH; ++c) {
      s += Qs[i * XB_LD + c] * Ks[j * XB_LD + c];
      dp += Gs[i * XB_LD + c] * Vs[j * XB_LD + c];
    }
    if (dr.thr) {  // the forward multiplied P by the dropout mask before P·V: same counter-based mask here
      const uint64_t idx = ((((uint64_t)(q0 + i)) * H + head) << 8) | (uint64_t)(j & 0xff);
      dp = stlt_keep(dr, site, idx) ? dp * dr.scale : 0.f;
    }
    Ps[i * PLD + j] = ok ? s * scale : -1e30f;
    Ds[i * PLD + j] = dp;
  }
  __syncthreads();
  if (tid < Lq) {
    const int i = tid;
    float m = -1e30f;
    for (int j = 0; j < Lk; ++j) m = fmaxf(m, Ps[i * PLD + j]);
    float l = 0.f;
    for (int j = 0; j < Lk; ++j) {
      const float sv = Ps[i * PLD + j];
      const float e = sv > -1e29f ? expf(sv - m) : 0.f;
      Ps[i * PLD + j] = e;
      l += e;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float dsum = 0.f;
    for (int j = 0; j < Lk; ++j) {
      const float pj = Ps[i * PLD + j] * inv;
      Ps[i * PLD + j] = pj;
      dsum += pj * Ds[i * PLD + j];
    }
    for (int j = 0; j < Lk; ++j) Ds[i * PLD + j] = Ps[i * PLD + j] * (Ds[i * PLD + j] - dsum);
  }
  __syncthreads();
  for (int idx = tid; idx < Lq * XB_DH; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    float acc = 0.f;
    for (int j = 0; j < Lk; ++j) acc += Ds[r * PLD + j] * Ks[j * XB_LD + c];
    dq[(q0 + r) * lddq + head * XB_DH + c] = acc * scale;
  }
  for (int idx = tid; idx < Lk * XB_DH; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < Lq; ++i) {
      ak += Ds[i * PLD + r] * Qs[i * XB_LD + c];
      float pd = Ps[i * PLD + r];
      if (dr.thr) {
        const uint64_t idx = ((((uint64_t)(q0 + i)) * H + head) << 8) | (uint64_t)(r & 0xff);
        pd = stlt_keep(dr, site, idx) ? pd * dr.scale : 0.f;
      }
      av += pd * Gs[i * XB_LD + c];
    }
    dk[(k0 + r) * lddkv + head * XB_DH + c] = ak * scale;
    dv[(k0 + r) * lddkv + head * XB_DH + c] = av;
  }
}

// The same backward for sequences of up to 256 tokens on either side (layouts of more than 64 frames through the fusion
// models): one block per (sequence, head), queries in tiles of 32, keys / values streamed twice per query tile through
// LDS (scores and dP, then dQ and the tile's share of dK / dV, which the thread that owns an output element adds up
// over the query tiles — a fixed order, no atomics).  Same FMA arithmetic and dropout indexing as the kernel above.
constexpr int XB_LONG_MAXL = 256;

__global__ __launch_bounds__(256) void attn_bwd_general_long_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                                    const float* __restrict__ v, int64_t ldkv,
                                                                    const float* __restrict__ dctx, const uint8_t* __restrict__ kpm,
                                                                    int causal, int Lq, int Lk, int H, float scale,
                                                                    float* __restrict__ dq, int64_t lddq, float* __restrict__ dk,
                                                                    float* __restrict__ dv, int64_t lddkv, StltDrop dr, uint32_t site) {
  extern __shared__ float xb_smem[];
  const int PLD = Lk + 1;
  float* Qs = xb_smem;                // 32 x 65: query tile
  float* Gs = Qs + 32 * XB_LD;        // dO tile
  float* Ks = Gs + 32 * XB_LD;        // key tile
  float* Vs = Ks + 32 * XB_LD;        // value tile
  float* Ps = Vs + 32 * XB_LD;        // 32 x (Lk+1): scores -> P
  float* Ds = Ps + 32 * PLD;          // dP -> dS
  const int tid = threadIdx.x;
  const int head = blockIdx.x % H;
  const int64_t sq = blockIdx.x / H;
  const int d = H * XB_DH;
  const int64_t qt0 = sq * Lq, kt0 = sq * Lk;
  const int rc = tid >> 6, c = tid & 63;  // this thread's row class (rows rc, rc+4, ...) and channel
  const float* qb = q + qt0 * ldq + head * XB_DH;
  const float* kb = k + kt0 * ldkv + head * XB_DH;
  const float* vb = v + kt0 * ldkv + head * XB_DH;
  float* dkb = dk + kt0 * lddkv + head * XB_DH;
  float* dvb = dv + kt0 * lddkv + head * XB_DH;
  for (int r = rc; r < Lk; r += 4) { dkb[r * lddkv + c] = 0.f; dvb[r * lddkv + c] = 0.f; }  // accumulated over the query tiles below
  for (int q0 = 0; q0 < Lq; q0 += 32) {
    const int nq = Lq - q0 < 32 ? Lq - q0 : 32;
    for (int i = rc; i < 32; i += 4) {
      const bool in = i < nq;
      Qs[i * XB_LD + c] = in ? qb[(q0 + i) * ldq + c] : 0.f;
      Gs[i * XB_LD + c] = in ? dctx[(qt0 + q0 + i) * (int64_t)d + head * XB_DH + c] : 0.f;
    }
    for (int k0 = 0; k0 < Lk; k0 += 32) {  // pass A: scores and dP against every key tile
      const int nkeys = Lk - k0 < 32 ? Lk - k0 : 32;
      __syncthreads();
      for (int j = rc; j < 32; j += 4) {
        const bool in = j < nkeys;
        Ks[j * XB_LD + c] = in ? kb[(k0 + j) * ldkv + c] : 0.f;
        Vs[j * XB_LD + c] = in ? vb[(k0 + j) * ldkv + c] : 0.f;
      }
      __syncthreads();
      for (int p = tid; p < 32 * 32; p += 256) {
        const int i = p >> 5, j = p & 31;
        const int kj = k0 + j, qi = q0 + i;
        const bool ok = i < nq && j < nkeys && (!kpm || kpm[kt0 + kj] == 0) && (!causal || kj <= qi);
        float sc = 0.f, dp = 0.f;
#pragma unroll 8
        for (int e = 0; e < XB_DH; ++e) {
          sc += Qs[i * XB_LD + e] * Ks[j * XB_LD + e];
          dp += Gs[i * XB_LD + e] * Vs[j * XB_LD + e];
        }
        if (dr.thr) {
          const uint64_t idx = ((((uint64_t)(qt0 + qi)) * H + head) << 8) | (uint64_t)(kj & 0xff);
          dp = stlt_keep(dr, site, idx) ? dp * dr.scale : 0.f;
        }
        if (j < nkeys) { Ps[i * PLD + kj] = ok ? sc * scale : -1e30f; Ds[i * PLD + kj] = dp; }
      }
    }
    __syncthreads();
    if (tid < 32) {  // row softmax, D_i, dS (in place: Ps <- P, Ds <- dS)
      const int i = tid;
      float m = -1e30f;
      for (int j = 0; j < Lk; ++j) m = fmaxf(m, Ps[i * PLD + j]);
      float l = 0.f;
      for (int j = 0; j < Lk; ++j) {
        const float sv = Ps[i * PLD + j];
        const float e = sv > -1e29f ? expf(sv - m) : 0.f;
        Ps[i * PLD + j] = e;
        l += e;
      }
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      float dsum = 0.f;
      for (int j = 0; j < Lk; ++j) {
        const float pj = Ps[i * PLD + j] * inv;
        Ps[i * PLD + j] = pj;
        dsum += pj * Ds[i * PLD + j];
      }
      for (int j = 0; j < Lk; ++j) Ds[i * PLD + j] = Ps[i * PLD + j] * (Ds[i * PLD + j] - dsum);
    }
    float dqa[8];  // pass B: dQ of this query tile, and its share of dK / dV
#pragma unroll
    for (int t = 0; t < 8; ++t) dqa[t] = 0.f;
    for (int k0 = 0; k0 < Lk; k0 += 32) {
      const int nkeys = Lk - k0 < 32 ? Lk - k0 : 32;
      __syncthreads();
      for (int j = rc; j < 32; j += 4) Ks[j * XB_LD + c] = j < nkeys ? kb[(k0 + j) * ldkv + c] : 0.f;
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int i = rc + 4 * t;
        float acc = 0.f;
        for (int j = 0; j < nkeys; ++j) acc += Ds[i * PLD + k0 + j] * Ks[j * XB_LD + c];
        dqa[t] += acc;
      }
      for (int j = rc; j < nkeys; j += 4) {
        const int kj = k0 + j;
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < nq; ++i) {
          ak += Ds[i * PLD + kj] * Qs[i * XB_LD + c];
          float pd = Ps[i * PLD + kj];
          if (dr.thr) {
            const uint64_t idx = ((((uint64_t)(qt0 + q0 + i)) * H + head) << 8) | (uint64_t)(kj & 0xff);
            pd = stlt_keep(dr, site, idx) ? pd * dr.scale : 0.f;
          }
          av += pd * Gs[i * XB_LD + c];
        }
        dkb[kj * lddkv + c] += ak * scale;  // the same thread owns (kj, c) in every query tile
        dvb[kj * lddkv + c] += av;
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = rc + 4 * t;
      if (i < nq) dq[(qt0 + q0 + i) * lddq + head * XB_DH + c] = dqa[t] * scale;
    }
    __syncthreads();  // the next query tile overwrites Qs / Gs / Ps / Ds
  }
}

// Element-wise pieces of the op-level training path: counter-based dropout (its own backward: the mask is a function of
// (seed, site, element index)) and the ReLU derivative taken from the activation's output.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, StltDrop dr, uint32_t site) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      *reinterpret_cast<f32x4*>(y + i) = stlt_drop4(dr, site, (uint64_t)i, *reinterpret_cast<const f32x4*>(x + i));
    } else {
      for (int64_t j = i; j < n; ++j) y[j] = stlt_keep(dr, site, (uint64_t)j) ? x[j] * dr.scale : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i), a = *reinterpret_cast<const f32x4*>(y + i);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = a[e] > 0.f ? g[e] : 0.f;
      *reinterpret_cast<f32x4*>(dx + i) = o;
    } else {
      for (int64_t j = i; j < n; ++j) dx[j] = y[j] > 0.f ? dy[j] : 0.f;
    }
  }
}

}  // namespace

extern "C" {

int stlt_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint32_t site, stlt_stream_t stream) {
  if (!x || !y) return stlt_set_error(STLT_EINVAL, "stlt_dropout: null pointer");
  if (!(p >= 0.f && p < 1.f)) return stlt_set_error(STLT_EINVAL, "dropout probability must be in [0,1)");
  if (n < 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return stlt_set_error(STLT_EINVAL, "stlt_dropout: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  StltProfScope ps(STLT_K_MISC, (hipStream_t)stream);
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, stlt_drop_make(p, seed), site);
  return stlt_check_launch("dropout_kernel");
}

int stlt_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, stlt_stream_t stream) {
  if (!dy || !y || !dx) return stlt_set_error(STLT_EINVAL, "stlt_relu_bwd: null pointer");
  if (n < 0 || ((uintptr_t)dy & 15) || ((uintptr_t)y & 15) || ((uintptr_t)dx & 15)) return stlt_set_error(STLT_EINVAL, "stlt_relu_bwd: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  StltProfScope ps(STLT_K_MISC, (hipStream_t)stream);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
  return stlt_check_launch("relu_bwd_kernel");
}

size_t stlt_linear_bwd_scratch_bytes(int64_t N) {
  return STLT_GEMM_SCRATCH_BYTES + (size_t)(N > 0 ? N : 0) * 64 * sizeof(float);
}

int stlt_linear_bwd(const float* x, const float* w, const float* dy, int64_t M, int64_t N, int64_t K, float* dx, float* dw, float* db,
                    stlt_ctx* ctx, void* scratch, size_t scratch_bytes, stlt_stream_t stream) {
  if (!x || !w || !dy || !scratch) return stlt_set_error(STLT_EINVAL, "stlt_linear_bwd: null pointer");
  if (M < 0 || N <= 0 || K <= 0) return stlt_set_error(STLT_EINVAL, "stlt_linear_bwd: bad shape");
  if (scratch_bytes < stlt_linear_bwd_scratch_bytes(N)) return stlt_set_error(STLT_EWORKSPACE, "stlt_linear_bwd: scratch %zu B < required %zu B", scratch_bytes, stlt_linear_bwd_scratch_bytes(N));
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  StltCtxScope ctx_scope(ctx, s);  // dx may be served from the context's transposed copy of w
  if (ctx_scope.error()) return ctx_scope.error();
  StltGemmScratch lend(scratch, STLT_GEMM_SCRATCH_BYTES);
  float* red = (float*)((char*)scratch + STLT_GEMM_SCRATCH_BYTES);
  const bool mfma_ok = N % 32 == 0 && K % 4 == 0;  // contraction lengths / leading dimensions the MFMA kernel takes
  if (dx) {  // dx (M,K) = dy (M,N) · W (N,K)
    bool small = false;
    if (mfma_ok) { if (int e = launch_input_grad_gemm16(dy, N, w, N, K, nullptr, 0, dx, K, M, s, &small)) return e; }
    if (small) {}
    else if (mfma_ok) { if (int e = launch_gemm(0, 1, dy, N, w, K, nullptr, nullptr, 0, dx, K, 0, M, K, N, 1, STLT_ACT_NONE, s)) return e; }
    else if (int e = launch_small_gemm(dy, N, 1, w, K, 1, dx, K, M, K, N, 0, s)) return e;
  }
  if (dw) {  // dw (N,K) += dyᵀ (N,M) · x (M,K): the MFMA kernel contracts over multiples of 32 rows, the rest goes to the strided kernel
    const int64_t Mf = (N % 4 == 0 && K % 4 == 0) ? M / 32 * 32 : 0;
    if (Mf > 0) { if (int e = launch_gemm(1, 1, dy, N, x, K, nullptr, dw, K, dw, K, 0, N, K, Mf, 1, STLT_ACT_NONE, s)) return e; }
    if (M > Mf) { if (int e = launch_small_gemm(dy + Mf * N, 1, N, x + Mf * K, K, 1, dw, K, N, K, M - Mf, 1, s)) return e; }
  }
  if (db) { if (int e = launch_colsum_acc(dy, N, M, N, db, red, s)) return e; }
  return 0;
}

int stlt_attn_fwd_dropout(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kpm, int causal,
                          int64_t S, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float dropout_p, uint64_t seed, uint32_t site, float* ctx,
                          stlt_stream_t stream) {
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return stlt_set_error(STLT_EINVAL, "dropout probability must be in [0,1)");
  return launch_attn_general(q, ldq, k, v, ldkv, kpm, causal, S, Lq, Lk, H, dh, ctx, causal ? STLT_K_ATTN_TEMPORAL : STLT_K_ATTN_SPATIAL,
                             (hipStream_t)stream, stlt_drop_make(dropout_p, seed), site);
}

int stlt_attn_bwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* dctx, const uint8_t* kpm,
                  int causal, int64_t S, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float dropout_p, uint64_t seed, uint32_t site,
                  float* dq, int64_t lddq, float* dk, float* dv, int64_t lddkv, stlt_stream_t stream) {
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return stlt_set_error(STLT_EINVAL, "dropout probability must be in [0,1)");
  if (!q || !k || !v || !dctx || !dq || !dk || !dv) return stlt_set_error(STLT_EINVAL, "stlt_attn_bwd: null pointer");
  if (dh != XB_DH) {  // any other head dim: attn_any.hip
    StltProfScope ps(STLT_K_ATTN_BWD, (hipStream_t)stream);
    return launch_attn_any_bwd(q, ldq, k, v, ldkv, dctx, kpm, nullptr, nullptr, nullptr, 0, causal, S, Lq, Lk, H, dh, dq, lddq, dk, dv, lddkv,
                               (hipStream_t)stream, stlt_drop_make(dropout_p, seed), site);
  }
  if (Lq <= 0 || Lk <= 0 || Lq > XB_LONG_MAXL || Lk > XB_LONG_MAXL) return stlt_set_error(STLT_EINVAL, "stlt_attn_bwd: sequences of at most %d tokens (got %lld / %lld)", XB_LONG_MAXL, (long long)Lq, (long long)Lk);
  if (causal && Lq != Lk) return stlt_set_error(STLT_EINVAL, "stlt_attn_bwd: causal masking needs Lq == Lk");
  if (Lq > XB_MAXL || Lk > XB_MAXL) {  // streamed variant: query tiles of 32, keys in tiles through LDS
    if (S == 0) return 0;
    if (S * H > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "stlt_attn_bwd: too many sequences");
    const size_t lds_long = ((size_t)4 * 32 * XB_LD + (size_t)2 * 32 * (Lk + 1)) * sizeof(float);
    static StltPerDeviceOnce long_once;
    bool& opted = long_once.flag();
    if (!opted) {
      if (hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_general_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); e != hipSuccess)
        return stlt_set_error((int)e, "stlt_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      opted = true;
    }
    StltProfScope ps(STLT_K_ATTN_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(attn_bwd_general_long_kernel, dim3((unsigned)(S * H)), dim3(256), lds_long, (hipStream_t)stream, q, ldq, k, v, ldkv, dctx, kpm,
                       causal, (int)Lq, (int)Lk, (int)H, 1.0f / sqrtf((float)dh), dq, lddq, dk, dv, lddkv, stlt_drop_make(dropout_p, seed), site);
    return stlt_check_launch("attn_bwd_general_long_kernel");
  }
  if (S == 0) return 0;
  if (S * H > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "stlt_attn_bwd: too many sequences");
  if (!causal) {  // at most 48 tokens on either side (the fusion models' 32 frames x 33 appearance tokens): the MFMA kernel of attn_bwdx16.hip
    bool taken = false;
    StltProfScope ps(STLT_K_ATTN_BWD, (hipStream_t)stream);
    const int rc = launch_attn_bwdx16(q, ldq, k, v, ldkv, dctx, kpm, S, Lq, Lk, H, dq, lddq, dk, dv, lddkv, stlt_drop_make(dropout_p, seed), site,
                                      (hipStream_t)stream, &taken);
    if (taken || rc != 0) return rc;
  }
  const size_t lds = ((size_t)2 * Lq * XB_LD + (size_t)2 * Lk * XB_LD + (size_t)2 * Lq * (Lk + 1)) * sizeof(float);
  static StltPerDeviceOnce once;  // a function attribute is set per device
  bool& opt_in = once.flag();
  if (!opt_in) {
    if (hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_general_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); e != hipSuccess)
      return stlt_set_error((int)e, "stlt_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    opt_in = true;
  }
  hipLaunchKernelGGL(attn_bwd_general_kernel, dim3((unsigned)(S * H)), dim3(256), lds, (hipStream_t)stream, q, ldq, k, v, ldkv, dctx, kpm, causal,
                     (int)Lq, (int)Lk, (int)H, 1.0f / sqrtf((float)dh), dq, lddq, dk, dv, lddkv, stlt_drop_make(dropout_p, seed), site);
  return stlt_check_launch("attn_bwd_general_kernel");
}

size_t stlt_attn_core_bwd_scratch_bytes(int64_t H) { return (size_t)256 * 3 * (size_t)(H > 0 ? H : 0) * 64 * sizeof(float); }

int stlt_attn_core_bwd(const float* qkv, const float* dctx, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, int64_t dh,
                       float dropout_p, uint64_t seed, uint32_t site, float* dqkv, float* in_proj_b_grad, void* scratch, size_t scratch_bytes,
                       stlt_stream_t stream) {
  if (!qkv || !dctx || !kpm || !dqkv) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_bwd: null pointer");
  if (dh < 1 || dh > 256 || H <= 0 || S < 0 || L <= 0) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_bwd: bad shape (head dim 1 ... 256)");
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return stlt_set_error(STLT_EINVAL, "dropout probability must be in [0,1)");
  if (in_proj_b_grad && (!scratch || scratch_bytes < stlt_attn_core_bwd_scratch_bytes(H))) return stlt_set_error(STLT_EWORKSPACE, "stlt_attn_core_bwd: scratch too small");
  if (S == 0) return 0;
  return launch_attn_bwd(qkv, dctx, kpm, causal, S, L, H, dh, dqkv, (hipStream_t)stream, stlt_drop_make(dropout_p, seed), site, in_proj_b_grad,
                         (float*)scratch);
}

size_t stlt_add_layernorm_bwd_scratch_bytes(int64_t d) { return (size_t)ln_bwd_scratch_floats(d > 0 ? d : 0) * sizeof(float); }

int stlt_add_layernorm_bwd(const float* dy, const float* x, const float* res, const float* ln_w, float eps, int64_t M, int64_t d,
                           float* ds, float* g_w, float* g_b, void* scratch, size_t scratch_bytes, stlt_stream_t stream) {
  if (scratch_bytes < stlt_add_layernorm_bwd_scratch_bytes(d)) return stlt_set_error(STLT_EWORKSPACE, "stlt_add_layernorm_bwd: scratch too small");
  return launch_ln_bwd(dy, d, x, d, res, d, ln_w, eps, M, d, ds, d, g_w, g_b, (float*)scratch, (hipStream_t)stream);
}

// K1 / K7 with their pre-LayerNorm sums kept (what their backward needs), and the parameter gradients of the sums.
int stlt_embed_fwd_train(const int64_t* categories, const float* boxes, const float* scores, const float* cat_table, int64_t n_categories,
                         const float* box_w, const float* box_b, const float* score_w, const float* score_b, const float* ln_w,
                         const float* ln_b, float eps, int64_t n_tokens, int64_t d, float* pre_out, float* out, stlt_stream_t stream) {
  if (!pre_out) return stlt_set_error(STLT_EINVAL, "stlt_embed_fwd_train: pre_out is null");
  return launch_embed(categories, boxes, scores, cat_table, n_categories, box_w, box_b, score_w, score_b, ln_w, ln_b, eps, n_tokens, d, out,
                      (hipStream_t)stream, pre_out);
}

size_t stlt_embed_bwd_scratch_bytes(int64_t n_tokens, int64_t n_categories, int64_t d) {
  return (size_t)embed_bwd_scratch_floats(n_tokens > 0 ? n_tokens : 0, n_categories > 0 ? n_categories : 1, d > 0 ? d : 0) * sizeof(float);
}

int stlt_embed_bwd(const float* d_pre, const int64_t* categories, const float* boxes, const float* scores, int64_t n_categories,
                   int64_t n_tokens, int64_t d, float* g_cat, float* g_box_w, float* g_box_b, float* g_score_w, float* g_score_b,
                   void* scratch, size_t scratch_bytes, stlt_stream_t stream) {
  if (scratch_bytes < stlt_embed_bwd_scratch_bytes(n_tokens, n_categories, d)) return stlt_set_error(STLT_EWORKSPACE, "stlt_embed_bwd: scratch too small");
  return launch_embed_bwd(d_pre, categories, boxes, scores, n_categories, n_tokens, d, g_cat, g_box_w, g_box_b, g_score_w, g_score_b,
                          (float*)scratch, (hipStream_t)stream);
}

int stlt_frames_embed_fwd_train(const float* spatial, int64_t row_stride, const int64_t* frame_types, const float* pos_table,
                                const float* type_table, const float* ln_w, const float* ln_b, float eps, int64_t B, int64_t T, int64_t d,
                                float* pre_out, float* out, stlt_stream_t stream) {
  if (!pre_out) return stlt_set_error(STLT_EINVAL, "stlt_frames_embed_fwd_train: pre_out is null");
  return launch_frames_embed(spatial, row_stride, frame_types, pos_table, type_table, ln_w, ln_b, eps, B, T, d, out, (hipStream_t)stream, pre_out);
}

size_t stlt_frames_embed_bwd_scratch_bytes(int64_t T, int64_t d) { return (size_t)16 * (size_t)((T > 0 ? T : 0) + 5) * (size_t)(d > 0 ? d : 0) * sizeof(float); }

int stlt_frames_embed_bwd(const float* d_pre, const int64_t* frame_types, int64_t B, int64_t T, int64_t d, float* g_pos, float* g_type,
                          void* scratch, size_t scratch_bytes, stlt_stream_t stream) {
  if (scratch_bytes < stlt_frames_embed_bwd_scratch_bytes(T, d)) return stlt_set_error(STLT_EWORKSPACE, "stlt_frames_embed_bwd: scratch too small");
  return launch_frames_bwd(d_pre, frame_types, B, T, 1, d, nullptr, g_pos, g_type, (float*)scratch, (hipStream_t)stream);
}

int stlt_gelu_fwd(const float* u, float* h, int64_t n, stlt_stream_t stream) { return launch_gelu_fwd(u, h, n, (hipStream_t)stream); }
int stlt_gelu_bwd(const float* dh, const float* u, float* du, int64_t n, stlt_stream_t stream) { return launch_gelu_bwd(dh, u, du, n, (hipStream_t)stream); }

}  // extern "C"
