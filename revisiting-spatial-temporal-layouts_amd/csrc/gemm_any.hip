// Products whose contraction length is not a multiple of 32 (hidden sizes like 100 or 200: the reference takes any
// hidden_size % num_attention_heads == 0, src/modelling/configs.py:92-111; every released checkpoint is 768).  The MFMA kernels of gemm.hip /
// gemm16.hip stage 32-wide k-slabs by LDS-DMA, which cannot mask a partial slab; this file is the cold path launch_gemm falls back to:
// a 64 x 64 tile per workgroup through LDS with bounds-checked loads, FMA on the vector ALU, the same epilogues (bias, exact-erf GELU /
// ReLU, add-source) and the same three operand layouts (forward x·Wᵀ, input gradient dY·W, weight gradient dYᵀ·X).  One fixed summation
// order per output element: bitwise reproducible.
#include "common.h"

namespace {

constexpr int GA_T = 64, GA_K = 16;

// C (M, N) = opA(A)·opB(B) [+ bias] [act] [+ R].  TA: A stored (K, M) (else (M, K)); TB: B stored (K, N) (else (N, K)).
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_any_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                       const float* __restrict__ bias, const float* r, int64_t ldr,
                                                       float* c, int64_t ldc, int M, int N, int K, int act) {  // r may alias c (accumulation)
  __shared__ float As[GA_K][GA_T + 4], Bs[GA_K][GA_T + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * GA_T, n0 = (int64_t)blockIdx.y * GA_T;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += GA_K) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;  // 1024 elements of each operand tile
      {
        const int mm = TA ? (idx & 63) : (idx >> 4), kk = TA ? (idx >> 6) : (idx & 15);
        const int64_t gm = m0 + mm;
        const int gk = k0 + kk;
        As[kk][mm] = (gm < M && gk < K) ? (TA ? a[(int64_t)gk * lda + gm] : a[gm * lda + gk]) : 0.f;
      }
      {
        const int nn = TB ? (idx & 63) : (idx >> 4), kk = TB ? (idx >> 6) : (idx & 15);
        const int64_t gn = n0 + nn;
        const int gk = k0 + kk;
        Bs[kk][nn] = (gn < N && gk < K) ? (TB ? b[(int64_t)gk * ldb + gn] : b[gn * ldb + gk]) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GA_K; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t gm = m0 + ty * 4 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t gn = n0 + tx * 4 + j;
      if (gn >= N) continue;
      float v = acc[i][j];
      if (bias) v += bias[gn];
      if (act == STLT_ACT_GELU) v = gelu_epilogue(v);
      else if (act == STLT_ACT_RELU) v = fmaxf(v, 0.f);
      if (r) v += r[gm * ldr + gn];
      c[gm * ldc + gn] = v;
    }
  }
}

}  // namespace

// launch_gemm's fallback (same operand conventions; no split, no fused GELU backward — the callers keep those for K % 32 == 0)
int launch_gemm_any(int transA, int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, const float* r,
                    int64_t ldr, float* c, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t s) {
  if (!a || !b || !c) return stlt_set_error(STLT_EINVAL, "gemm: null pointer");
  if (M < 0 || N <= 0 || K <= 0 || M > 0x7fffff00LL || N > 0x7fffff00LL || K > 0x7fffff00LL) return stlt_set_error(STLT_EINVAL, "gemm: bad shape");
  if (transA && !transB) return stlt_set_error(STLT_EINVAL, "gemm: the (transA, !transB) layout is not built");
  if (act != STLT_ACT_NONE && act != STLT_ACT_GELU && act != STLT_ACT_RELU) return stlt_set_error(STLT_EINVAL, "gemm: activation %d needs a contraction length that is a multiple of 32", act);
  if ((r && ldr < N) || ldc < N || lda < (transA ? M : K) || ldb < (transB ? N : K)) return stlt_set_error(STLT_EINVAL, "gemm: bad leading dimension");
  if (M == 0) return 0;
  const int64_t gy = (M + GA_T - 1) / GA_T, gx = (N + GA_T - 1) / GA_T;
  if (gx > 65535) return stlt_set_error(STLT_EINVAL, "gemm: too many columns for a contraction length that is not a multiple of 32 (N=%lld)", (long long)N);
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)M * (double)N * (double)K);
  const dim3 grid((unsigned)gy, (unsigned)gx), block(256);
#define GA_LAUNCH(TAV, TBV) hipLaunchKernelGGL((gemm_any_kernel<TAV, TBV>), grid, block, 0, s, a, lda, b, ldb, bias, r, ldr, c, ldc, (int)M, (int)N, (int)K, act)
  if (transA) GA_LAUNCH(true, true);
  else if (transB) GA_LAUNCH(false, true);
  else GA_LAUNCH(false, false);
#undef GA_LAUNCH
  return stlt_check_launch("gemm_any_kernel");
}
