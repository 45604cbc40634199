// Products whose contraction length is not a multiple of 32 (hidden sizes like 100 or 200: the reference takes any
// hidden_size % num_attention_heads == 0, src/modelling/configs.py:92-111; every released checkpoint is 768).  The MFMA kernels of gemm.hip /
// gemm16.hip stage 32-wide k-slabs by LDS-DMA, which cannot mask a partial slab; this file is the cold path launch_gemm falls back to:
// a 64 x 64 tile per workgroup through LDS with bounds-checked (zero-filled) loads, v_mfma_f32_16x16x4_f32 on the LDS image, the same epilogues (bias, exact-erf GELU /
// ReLU, add-source) and the same three operand layouts (forward x·Wᵀ, input gradient dY·W, weight gradient dYᵀ·X).  One fixed summation
// order per output element: bitwise reproducible.
#include <cstdint>
#include "common.h"

namespace {

constexpr int GA_T = 64, GA_K = 32;

// C (M, N) = opA(A)·opB(B) [+ bias] [act] [+ R].  TA: A stored (K, M) (else (M, K)); TB: B stored (K, N) (else (N, K)).
// 4 waves, each a 32 x 32 quadrant of the tile = 2 x 2 v_mfma_f32_16x16x4_f32 blocks: lane (li = lane & 15, lg = lane >> 4) feeds
// A[row li][k lg] and B[k lg][col li] of a block and holds D[rows 4 lg .. 4 lg + 3][col li].
// four consecutive floats at p, of which the first `n` (0 ... 4) exist; VEC: p is 16-byte aligned
template <bool VEC>
__device__ __forceinline__ f32x4 load4_guarded(const float* p, int n) {
  if (n >= 4 && VEC) return *reinterpret_cast<const f32x4*>(p);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n > 0) v[0] = p[0];
  if (n > 1) v[1] = p[1];
  if (n > 2) v[2] = p[2];
  if (n > 3) v[3] = p[3];
  return v;
}
// one operand tile into its k-major LDS image img[k][row] (zeros beyond the matrix and beyond K).  T = false: the operand is stored
// [row][k] — a lane takes four consecutive k of one row (lanes = consecutive rows: conflict-free LDS writes; a row's 128-byte line is
// used up by the eight lane groups); T = true: stored [k][row] — four consecutive rows of one k, one 16-byte LDS write.
template <bool T, bool VEC>
__device__ __forceinline__ void stage_tile(float (*img)[GA_T + 4], const float* __restrict__ src, int64_t ld, int64_t row0, int rows, int k0, int K, int tid) {
#pragma unroll
  for (int e = 0; e < (GA_T * GA_K / 4) / 256; ++e) {
    const int idx = tid + 256 * e;
    if (!T) {
      const int rr = idx & 63, kq = (idx >> 6) * 4;
      const int64_t g = row0 + rr;
      const int left = K - (k0 + kq);
      const f32x4 v = g < rows ? load4_guarded<VEC>(src + g * ld + k0 + kq, left < 0 ? 0 : left) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) img[kq + j][rr] = v[j];
    } else {
      const int kk = idx >> 4, rq = (idx & 15) * 4;
      const int64_t left = (int64_t)rows - (row0 + rq);
      const f32x4 v = k0 + kk < K ? load4_guarded<VEC>(src + (int64_t)(k0 + kk) * ld + row0 + rq, left < 0 ? 0 : (left > 4 ? 4 : (int)left)) : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&img[kk][rq]) = v;
    }
  }
}

// SPLIT: blockIdx.z takes k-slabs [z * slabs_per_split, ...) and stores its raw partial tile into c = part + z * (rows rounded up to 64) * N
// (ldc = N; no bias, activation or add-source: skinny_finish_kernel applies them to the sum of the partials in a fixed order).
template <bool TA, bool TB, bool VEC, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_any_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                       const float* __restrict__ bias, const float* r, int64_t ldr,
                                                       float* c, int64_t ldc, int M, int N, int K, int act, int slabs_per_split = 0) {  // r may alias c (accumulation)
  __shared__ __attribute__((aligned(16))) float As[GA_K][GA_T + 4], Bs[GA_K][GA_T + 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int64_t m0 = (int64_t)blockIdx.x * GA_T, n0 = (int64_t)blockIdx.y * GA_T;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int k_begin = SPLIT ? (int)blockIdx.z * slabs_per_split * GA_K : 0;
  const int k_end = SPLIT ? (k_begin + slabs_per_split * GA_K < K ? k_begin + slabs_per_split * GA_K : K) : K;
  if (SPLIT) c += (int64_t)blockIdx.z * (((int64_t)M + GA_T - 1) / GA_T * GA_T) * (int64_t)N;
  for (int k0 = k_begin; k0 < k_end; k0 += GA_K) {
    stage_tile<TA, VEC>(As, a, lda, m0, M, k0, K, tid);
    stage_tile<TB, VEC>(Bs, b, ldb, n0, N, k0, K, tid);
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < GA_K / 4; ++k4) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = As[4 * k4 + lg][wm + 16 * i + li];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = Bs[4 * k4 + lg][wn + 16 * j + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t gn = n0 + wn + 16 * j + li;
      if (gn >= N) continue;
      const float bn = (!SPLIT && bias) ? bias[gn] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t gm = m0 + wm + 16 * i + 4 * lg + q;
        if (gm >= M) continue;
        if (SPLIT) { c[gm * ldc + gn] = acc[i][j][q]; continue; }
        float v = acc[i][j][q] + bn;
        if (act == STLT_ACT_GELU) v = gelu_epilogue(v);
        else if (act == STLT_ACT_RELU) v = fmaxf(v, 0.f);
        if (r) v += r[gm * ldr + gn];
        c[gm * ldc + gn] = v;
      }
    }
}

// y[m][n] = act(bias[n] + part[0][m][n] + part[1][m][n] + ...) (+ r[m][n]): the partial tiles of a split product, summed in split order
__global__ __launch_bounds__(256) void skinny_finish_kernel(const float* __restrict__ part, int64_t split_stride, int splits, const float* __restrict__ bias,
                                                            const float* r, int64_t ldr, float* y, int64_t ldy, int M, int N, int act) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
  float v = bias ? bias[n] : 0.f;
  const float* p = part + idx;
  for (int z = 0; z < splits; ++z) v += p[(int64_t)z * split_stride];
  if (act == STLT_ACT_GELU) v = gelu_epilogue(v);
  else if (act == STLT_ACT_RELU) v = fmaxf(v, 0.f);
  if (r) v += r[(int64_t)m * ldr + n];
  y[(int64_t)m * ldy + n] = v;
}

}  // namespace

// Products with at most 128 output rows (the rows a 64-clip batch reads from the last layer of a tower — one per clip —, the prediction head):
// whole tiles leave the chip empty (12 workgroups of 24 serial k-steps for 64 x 768 x 768: 18 us; 49 us on the large kernel's stream-K for
// K = 3072), so the contraction is split over ~384 workgroups of 64 x 64 partial tiles and a second small launch sums the partials in split
// order and applies bias / activation / add-source.  Deterministic; the partials live in the lent stream-K scratch.  *taken = false: not
// this path's (more rows, no scratch lent, switched off with STLT_GEMM_SKINNY=0).
int launch_gemm_skinny(int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, const float* r, int64_t ldr, float* c,
                       int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t s, bool* taken) {
  *taken = false;
  static const bool on = [] { const char* e = getenv("STLT_GEMM_SKINNY"); return !(e && e[0] == '0'); }();
  static const int max_rows = [] { const char* e = getenv("STLT_GEMM_SKINNY_ROWS"); return e ? atoi(e) : 128; }();
  if (!on || M <= 0 || M > max_rows || N <= 0 || K < 2 * GA_K || !a || !b || !c) return 0;
  if (act != STLT_ACT_NONE && act != STLT_ACT_GELU && act != STLT_ACT_RELU) return 0;
  if (N > 0x3fffffLL || K > 0x3fffff00LL || (r && ldr < N) || ldc < N || lda < K || ldb < (transB ? N : K)) return 0;
  size_t lent = 0;
  float* part = stlt_gemm_scratch_ptr(&lent);
  const int64_t gy = (M + GA_T - 1) / GA_T, gx = (N + GA_T - 1) / GA_T, slabs = (K + GA_K - 1) / GA_K;
  int64_t splits = (384 + gx * gy - 1) / (gx * gy);
  if (splits > slabs) splits = slabs;
  if (splits < 2) return 0;  // already enough tiles: the whole-tile kernels are the better fit
  const int64_t per = (slabs + splits - 1) / splits;
  splits = (slabs + per - 1) / per;
  const int64_t split_stride = gy * GA_T * N;
  if (!part || lent < (size_t)(splits * split_stride) * sizeof(float) || gx > 65535 || splits > 65535) return 0;
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)M * (double)N * (double)K);
  stlt_prof_note("gemm(skinny%s) M=%lld N=%lld K=%lld act=%d%s tile=64x64 tiles=%lld splits=%lld wg=%lld ksteps=%lld (+finish)", transB ? ", dX" : "", (long long)M, (long long)N,
                 (long long)K, act, r ? "+R" : "", (long long)(gx * gy), (long long)splits, (long long)(gx * gy * splits), (long long)per);
  const dim3 grid((unsigned)gy, (unsigned)gx, (unsigned)splits), block(256);
  const bool vec = lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0;
#define GS_LAUNCH(TBV) do { if (vec) hipLaunchKernelGGL((gemm_any_kernel<false, TBV, true, true>), grid, block, 0, s, a, lda, b, ldb, nullptr, nullptr, 0, part, N, (int)M, (int)N, (int)K, 0, (int)per); \
  else hipLaunchKernelGGL((gemm_any_kernel<false, TBV, false, true>), grid, block, 0, s, a, lda, b, ldb, nullptr, nullptr, 0, part, N, (int)M, (int)N, (int)K, 0, (int)per); } while (0)
  if (transB) GS_LAUNCH(true); else GS_LAUNCH(false);
#undef GS_LAUNCH
  *taken = true;
  if (int e = stlt_check_launch("gemm_any_kernel(split)")) return e;
  hipLaunchKernelGGL(skinny_finish_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, part, split_stride, (int)splits, bias, r, ldr, c, ldc, (int)M, (int)N, act);
  return stlt_check_launch("skinny_finish_kernel");
}

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
  stlt_prof_note("gemm(any%s) M=%lld N=%lld K=%lld act=%d%s tile=64x64 tiles=%lld wg=%lld ksteps=%lld", transA ? ", dW" : (transB ? ", dX" : ""), (long long)M, (long long)N, (long long)K, act,
                 r ? "+R" : "", (long long)(gx * gy), (long long)(gx * gy), (long long)((K + GA_K - 1) / GA_K));
  const dim3 grid((unsigned)gy, (unsigned)gx), block(256);
  // 16-byte loads where every row start is 16-byte aligned (pitches that are multiples of 4 floats: hidden sizes that are multiples of 4)
  const bool vec = lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0;
#define GA_LAUNCH(TAV, TBV) do { if (vec) hipLaunchKernelGGL((gemm_any_kernel<TAV, TBV, true>), grid, block, 0, s, a, lda, b, ldb, bias, r, ldr, c, ldc, (int)M, (int)N, (int)K, act); \
  else hipLaunchKernelGGL((gemm_any_kernel<TAV, TBV, false>), grid, block, 0, s, a, lda, b, ldb, bias, r, ldr, c, ldc, (int)M, (int)N, (int)K, act); } while (0)
  if (transA) GA_LAUNCH(true, true);
  else if (transB) GA_LAUNCH(false, true);
  else GA_LAUNCH(false, false);
#undef GA_LAUNCH
  return stlt_check_launch("gemm_any_kernel");
}
