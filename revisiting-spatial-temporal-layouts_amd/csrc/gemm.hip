// nn.Linear forward  Y = act(X · Wᵀ + b)  on the f32-input matrix cores of gfx950.
//
//   X (M,K) row-major with leading dim ldx, W (N,K) row-major (torch (out,in)), Y (M,N) with ldy.
//   Both operands are K-contiguous, so A- and B-fragments have the same shape: lane (r = lane&31,
//   h = lane>>5) of a wave reads 16 B = 4 consecutive k of row r (ds_read_b128) and feeds them to four
//   v_mfma_f32_32x32x2_f32.  The k index inside an 8-wide chunk is permuted (MFMA e pairs k = 8c+e with
//   k = 8c+4+e); a sum over k does not care, and A and B use the same permutation.
//
//   Block tile 128x128x32, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator
//   VGPRs).  LDS rows are padded to 36 floats: conflict-free for the ds_read_b128 lane groups.
//   Global->LDS staging goes through registers (issue the next tile's loads before the MFMAs of the
//   current one, write them to the other LDS buffer after), one barrier per k-tile, 2 blocks per CU.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;
constexpr int GEMM_THREADS = 256;

template <int ACT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_kernel(const float* __restrict__ X, int64_t ldx,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ Y, int64_t ldy, int M, int N,
                                                                  int K, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_LD];
  constexpr int STAGE = (BM + BN) * LDS_LD;  // floats per pipeline stage: A tile then B tile

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of tiles, walked N-fastest: neighbours then share the X panel and stream W through L2.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  // staging map: float4 f = tid + 256*i -> row f/8, 16-B column f%8 ; 8 consecutive lanes read one 128-B line
  const int srow = tid >> 3, scol = (tid & 7) * 4;
  const float* xg[4];
  const float* wg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int rm = m0 + srow + 32 * i;
    rm = rm < M ? rm : M - 1;  // clamp: ragged tiles re-read the last row, stores are guarded
    xg[i] = X + (int64_t)rm * ldx + scol;
    int rn = n0 + srow + 32 * i;
    rn = rn < N ? rn : N - 1;
    wg[i] = W + (int64_t)rn * K + scol;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f32x4 xa[4], wb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xa[i] = *reinterpret_cast<const f32x4*>(xg[i]);
    wb[i] = *reinterpret_cast<const f32x4*>(wg[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<f32x4*>(&smem[(srow + 32 * i) * LDS_LD + scol]) = xa[i];
    *reinterpret_cast<f32x4*>(&smem[(BM + srow + 32 * i) * LDS_LD + scol]) = wb[i];
  }
  __syncthreads();

  const int nk = K / BK;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      const int ko = (kt + 1) * BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xa[i] = *reinterpret_cast<const f32x4*>(xg[i] + ko);
        wb[i] = *reinterpret_cast<const f32x4*>(wg[i] + ko);
      }
    }
    const float* Ac = smem + cur * STAGE + (wm * 64 + lr) * LDS_LD + 4 * lh;
    const float* Bc = smem + cur * STAGE + (BM + wn * 64 + lr) * LDS_LD + 4 * lh;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      f32x4 a0 = *reinterpret_cast<const f32x4*>(Ac + 8 * c);
      f32x4 a1 = *reinterpret_cast<const f32x4*>(Ac + 32 * LDS_LD + 8 * c);
      f32x4 b0 = *reinterpret_cast<const f32x4*>(Bc + 8 * c);
      f32x4 b1 = *reinterpret_cast<const f32x4*>(Bc + 32 * LDS_LD + 8 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
      }
    }
    if (more) {
      float* An = smem + (cur ^ 1) * STAGE;
      float* Bn = An + BM * LDS_LD;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(&An[(srow + 32 * i) * LDS_LD + scol]) = xa[i];
        *reinterpret_cast<f32x4*>(&Bn[(srow + 32 * i) * LDS_LD + scol]) = wb[i];
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int n = n0 + wn * 64 + b * 32 + lr;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int mb = m0 + wm * 64 + a * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m < M) {
          float v = acc[a][b][r] + bv;
          if (ACT == STLT_ACT_GELU) v = gelu_erf(v);
          Y[(int64_t)m * ldy + n] = v;
        }
      }
    }
  }
}

}  // namespace

int launch_linear(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int64_t M,
                  int64_t N, int64_t K, int act, hipStream_t s) {
  if (!x || !w || !y) return stlt_set_error(STLT_EINVAL, "stlt_linear_fwd: null pointer");
  if (M < 0 || N <= 0 || K <= 0 || K % BK != 0)
    return stlt_set_error(STLT_EINVAL, "stlt_linear_fwd: K=%lld must be a positive multiple of %d (N=%lld)", (long long)K, BK, (long long)N);
  if (ldx % 4 != 0 || ldx < K || ldy < N)
    return stlt_set_error(STLT_EINVAL, "stlt_linear_fwd: ldx=%lld must be a multiple of 4 and >= K, ldy=%lld >= N", (long long)ldx, (long long)ldy);
  if (M > 0x7fffff00LL || N > 0x7fffff00LL) return stlt_set_error(STLT_EINVAL, "stlt_linear_fwd: M/N too large");
  if (act != STLT_ACT_NONE && act != STLT_ACT_GELU) return stlt_set_error(STLT_EINVAL, "stlt_linear_fwd: unknown activation %d", act);
  if (M == 0) return 0;
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
  StltProfScope ps(STLT_K_GEMM, s);
  dim3 grid((unsigned)(tiles_m * tiles_n)), block(GEMM_THREADS);
  if (act == STLT_ACT_GELU)
    hipLaunchKernelGGL((gemm_nt_kernel<STLT_ACT_GELU>), grid, block, 0, s, x, ldx, w, bias, y, ldy, (int)M, (int)N, (int)K, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<STLT_ACT_NONE>), grid, block, 0, s, x, ldx, w, bias, y, ldy, (int)M, (int)N, (int)K, tiles_m, tiles_n);
  return stlt_check_launch("gemm_nt_kernel");
}
