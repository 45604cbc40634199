// nn.Linear forward  Y = act(X·Wᵀ + b) (+ R)  with f32 operands and an f32-equivalent result on the BF16 matrix cores
// ("split-bf16", opt-in: STLT_GEMM_SPLIT_BF16=6 / stlt_set_gemm_split_bf16; never the default and never bench.py's `value`).
//
// Every f32 operand element is cut, in registers, into three bf16 pieces  a = a0 + a1 + a2  (a0 = the upper 16 bits of a,
// a1 = the upper 16 bits of a - a0, a2 = a - a0 - a1 rounded to bf16: 24 significand bits in all, the subtractions are exact)
// and a product a·b is taken as the six piece products whose weight is at least 2^-24 of it,
//     a0·b0 + a0·b1 + a1·b0 + a1·b1 + a0·b2 + a2·b0,
// each a v_mfma_f32_32x32x16_bf16 accumulating in f32: products of bf16 pieces are exact in f32, the dropped terms (a1·b2,
// a2·b1, a2·b2) are below 2^-24 |a·b| — the rounding an f32 multiply makes anyway.  The bf16 MFMA issues 16x the FLOPs of the
// f32 one per cycle, six of them make one f32-equivalent product: 2.7x on paper.  What bounds the kernel instead is the
// operand fetch: the chip moves ≈ 6.4 TB/s from L2 into LDS (MI355X_MICROARCH.md: ldsdma-fill), a 256 x 128 tile of f32
// operands needs 0.0234 B per FLOP, so ≈ 273 TFLOP/s = 1.74x the f32-MFMA peak is the ceiling of this tile shape whatever the
// matrix pipe does (pre-split bf16 planes in memory would need 1.5x the bytes: worse).
//
// Structure = gemm.hip's loader-wave build (256 x 128 x 32 tiles, 8 MFMA waves as 4 x 2 of 64 x 64, 4 DMA-only loader waves
// two k-steps ahead with a counted vmcnt, three 48-KB LDS stages with the source-side bank swizzle, one barrier per k-step,
// persistent workgroups walking an XCD-contiguous band-major tile order, bias as the accumulators' initial value, transposed
// tile for 16-byte stores).  Forward (NT) layout only, whole-tile launches only (no stream-K): shapes the launcher does not
// take fall back to gemm.hip.
#include <cstdlib>
#include "common.h"

#ifndef STLT_X3_EXP
#define STLT_X3_EXP 0
#endif
#ifndef STLT_X3_SCHED
#define STLT_X3_SCHED 0  // > 0: pin that many VALU issues behind every MFMA of a phase (sched_group_barrier)
#endif

namespace {

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int X_WAVES = 8, X_LOADERS = 4;
constexpr int X_THREADS = 64 * (X_WAVES + X_LOADERS);
constexpr int NSTAGE = 3;
constexpr int STAGE_FLOATS = (BM + BN) * BK;

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_bfree(float x) {  // the branch-free erf GELU of gemm.hip's epilogue
  const float z = x * 0.70710678118654752440f;
  const float t = fminf(fabsf(z), 3.95f);
  float q = 1.1830035617776957e-07f;
  q = fmaf(q, t, -3.0875787615514128e-06f);
  q = fmaf(q, t, 3.5860794014297426e-05f);
  q = fmaf(q, t, -0.00024206875241361558f);
  q = fmaf(q, t, 0.0010191010078415275f);
  q = fmaf(q, t, -0.002435620641335845f);
  q = fmaf(q, t, 0.00011764218652388081f);
  q = fmaf(q, t, 0.027792135253548622f);
  q = fmaf(q, t, -0.14836618304252625f);
  q = fmaf(q, t, -0.9184255599975586f);
  q = fmaf(q, t, -1.6279090642929077f);
  q = fmaf(q, t, 2.831300349726007e-08f);
  const float e = copysignf(1.0f - __builtin_amdgcn_exp2f(q), z);
  return 0.5f * x * (1.0f + e);
}

// eight f32 (two 16-byte LDS reads) -> three planes of eight bf16 (4 dwords each): p0 | p1 = upper halves, p2 rounded
struct Planes { u32x4 p0, p1, p2; };
__device__ __forceinline__ Planes split8(f32x4 lo, f32x4 hi) {
  Planes o;
#if STLT_X3_EXP & 1  // timing experiment: no VALU work in the cut (results are wrong)
  o.p0 = __builtin_bit_cast(u32x4, lo); o.p1 = __builtin_bit_cast(u32x4, hi); o.p2 = o.p0;
  return o;
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = i < 2 ? lo[2 * i] : hi[2 * i - 4], x1 = i < 2 ? lo[2 * i + 1] : hi[2 * i - 3];
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    o.p0[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // (u1 & 0xffff0000) | (u0 >> 16)
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    o.p1[i] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 t = {(__bf16)s0, (__bf16)s1};  // v_cvt_pk_bf16_f32, round to nearest even
    o.p2[i] = __builtin_bit_cast(unsigned, t);
  }
  return o;
}
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int ACT, bool ADD>
__global__ __launch_bounds__(X_THREADS, 3) void gemm_nt_bf16x3_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                                     int64_t ldw, const float* __restrict__ bias, const float* __restrict__ R,
                                                                     int64_t ldr, float* __restrict__ Y, int64_t ldy, int M, int N, int K,
                                                                     int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE_FLOATS + 2 * BN];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nk = K / BK;
  const int n_tiles = tiles_m * tiles_n;
  // band-major, XCD-contiguous tile order (gemm.hip): XCD x walks positions [x*R*Gx, (x+1)*R*Gx)
  constexpr int GROUP_M = 4;
  const int G = gridDim.x;
  const int g_xcd = blockIdx.x & 7, g_local = blockIdx.x >> 3, g_gx = G >> 3;
  const int g_rounds = (n_tiles + G - 1) / G;
  int my_tiles = 0;
  {
    const int p0 = g_xcd * g_rounds * g_gx + g_local;
    if (p0 < n_tiles) my_tiles = (n_tiles - p0 + g_gx - 1) / g_gx;
    if (my_tiles > g_rounds) my_tiles = g_rounds;
  }
  if (my_tiles <= 0) return;
  const int total_steps = my_tiles * nk;
  auto tile_origin = [&](int it, int& m0, int& n0) {
    const int p = (g_xcd * g_rounds + it) * g_gx + g_local;
    const int band = p / (GROUP_M * tiles_n), w = p - band * (GROUP_M * tiles_n);
    const int rows = tiles_m - band * GROUP_M < GROUP_M ? tiles_m - band * GROUP_M : GROUP_M;
    const int tn_g = w / rows;
    m0 = (band * GROUP_M + (w - tn_g * rows)) * BM;
    n0 = tn_g * BN;
  };
  float* bias_lds = smem + NSTAGE * STAGE_FLOATS;

  if (wave >= X_WAVES) {
    // ---- loader waves: loader L issues A rows [64L, 64L+64) (8 instructions) and B rows [32L, 32L+32) (4) per k-step
    const int L = wave - X_WAVES;
    const int drow = lane >> 3, dslot = lane & 7;
    const float* pa[8];
    const float* pb[4];
    auto set_tile = [&](int it) {
      int m0, n0;
      tile_origin(it, m0, n0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = L * 64 + i * 8 + drow;
        int gm = m0 + r;
        gm = gm < M ? gm : M - 1;
        pa[i] = X + (int64_t)gm * ldx + (dslot ^ ((r >> 1) & 7)) * 4;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = L * 32 + i * 8 + drow;
        int gn = n0 + r;
        gn = gn < N ? gn : N - 1;
        pb[i] = W + (int64_t)gn * ldw + (dslot ^ ((r >> 1) & 7)) * 4;
      }
    };
    auto dma_bias = [&](int it) {
      if (bias && L == 0) {
        int m0, n0;
        tile_origin(it, m0, n0);
        float* dst = bias_lds + (it & 1) * BN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int n = n0 + i * 64 + lane;
          n = n < N ? n : N - 1;
          __builtin_amdgcn_global_load_lds((glb_void_ptr)(bias + n), (lds_void_ptr)(dst + i * 64), 4, 0, 0);
        }
      }
    };
    int l_it = 0, l_kt = 0, l_stage = 0;
    auto l_step = [&]() {
      if (l_kt == 0) set_tile(l_it);
      float* sa = smem + l_stage * STAGE_FLOATS + (L * 64) * BK;
      float* sb = smem + l_stage * STAGE_FLOATS + BM * BK + (L * 32) * BK;
#pragma unroll
      for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((glb_void_ptr)(pa[i] + l_kt * BK), (lds_void_ptr)(sa + i * 8 * BK), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_void_ptr)(pb[i] + l_kt * BK), (lds_void_ptr)(sb + i * 8 * BK), 16, 0, 0);
      if (++l_kt == nk) { ++l_it; l_kt = 0; }
      if (++l_stage == NSTAGE) l_stage = 0;
    };
    dma_bias(0);
    l_step();
    if (total_steps > 1) {
      l_step();
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int w_it = 0, w_kt = 0;
    for (int step = 0; step < total_steps; ++step) {
      if (w_kt == nk - 1 && w_it + 1 < my_tiles) dma_bias(w_it + 1);
      if (step + 2 < total_steps) {
        l_step();
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (++w_kt == nk) { ++w_it; w_kt = 0; }
    }
    return;
  }

  // ---- MFMA waves: 64 x 64 per wave (2 x 2 tiles of 32 x 32), transposed (D[n][m])
  const int wm = wave >> 1, wn = wave & 1;
  const int sw = (lr >> 1) & 7;
  const int a_row = (wm * 64 + lr) * BK;
  const int b_row = (BM + wn * 64 + lr) * BK;
  f32x16 acc[2][2];
  auto init_acc = [&](int it) {
    if (bias) {
      const float* src = bias_lds + (it & 1) * BN + wn * 64 + 4 * lh;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + b * 32 + 8 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[0][b][4 * q + j] = v[j]; acc[1][b][4 * q + j] = v[j]; }
        }
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
  };
  // raw f32 fragment of one 32-row operand block and one 16-wide k block: lane (r, h) holds k = 16 kb + 8 h .. + 7 of its row
  // as two 16-byte chunks
  struct Frag { f32x4 lo, hi; };
  constexpr int A0 = 0, A1 = 32 * BK;
  const int b_rel = b_row - a_row;
  auto read_frag = [&](int stage, int kb, int off) {
    const float* s = smem + stage * STAGE_FLOATS + a_row + off;
    Frag f;
    f.lo = *reinterpret_cast<const f32x4*>(s + ((4 * kb + 2 * lh) ^ sw) * 4);
    f.hi = *reinterpret_cast<const f32x4*>(s + ((4 * kb + 2 * lh + 1) ^ sw) * 4);
    return f;
  };
  auto six = [&](f32x16& d, const Planes& w, const Planes& x) {  // W pieces as the A operand, X pieces as B: D[n][m]; small terms first
#if !(STLT_X3_EXP & 2)  // timing experiment: three products only (results are wrong)
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w.p2), as_bf(x.p0), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w.p0), as_bf(x.p2), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w.p1), as_bf(x.p1), d, 0, 0, 0);
#endif
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w.p1), as_bf(x.p0), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w.p0), as_bf(x.p1), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w.p0), as_bf(x.p0), d, 0, 0, 0);
  };
  // one phase = the six products of one 32 x 32 block; the VALU work of cutting the fragment a later phase needs is written
  // next to it so that it issues in the MFMA gaps (6 issue slots of 4 cycles per 32-cycle MFMA: MI355X_MICROARCH.md)
#if STLT_X3_SCHED
#define X3_PHASE() do { for (int i_ = 0; i_ < 6; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, STLT_X3_SCHED, 0); } } while (0)
#else
#define X3_PHASE() do { } while (0)
#endif

  __builtin_amdgcn_s_barrier();
  init_acc(0);
  int c_it = 0, c_kt = 0, stage = 0;
  Planes XA0, XA1, WB0, WB1;
  Frag rWB1, rXA1;
  {
    const Frag a = read_frag(0, 0, A0), b = read_frag(0, 0, b_rel);
    rWB1 = read_frag(0, 0, b_rel + A1);
    rXA1 = read_frag(0, 0, A1);
    XA0 = split8(a.lo, a.hi);
    WB0 = split8(b.lo, b.hi);
  }
  for (int step = 0; step < total_steps; ++step) {
    const int next_stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    // ---- k block 0 of the stage (the fragments of k block 1 are read and cut underneath)
    Frag rXA0n = read_frag(stage, 1, A0), rWB0n = read_frag(stage, 1, b_rel);
    six(acc[0][0], WB0, XA0);
    WB1 = split8(rWB1.lo, rWB1.hi);
    X3_PHASE();
    rWB1 = read_frag(stage, 1, b_rel + A1);
    Frag rXA1n = read_frag(stage, 1, A1);
    six(acc[0][1], WB1, XA0);
    XA1 = split8(rXA1.lo, rXA1.hi);
    X3_PHASE();
    six(acc[1][0], WB0, XA1);
    XA0 = split8(rXA0n.lo, rXA0n.hi);
    X3_PHASE();
    six(acc[1][1], WB1, XA1);
    WB0 = split8(rWB0n.lo, rWB0n.hi);
    X3_PHASE();
    // ---- k block 1; every read of this stage has been issued: retire it, the next stage has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rXA0n = read_frag(next_stage, 0, A0);
    rWB0n = read_frag(next_stage, 0, b_rel);
    six(acc[0][0], WB0, XA0);
    WB1 = split8(rWB1.lo, rWB1.hi);
    X3_PHASE();
    rWB1 = read_frag(next_stage, 0, b_rel + A1);
    rXA1 = read_frag(next_stage, 0, A1);
    six(acc[0][1], WB1, XA0);
    XA1 = split8(rXA1n.lo, rXA1n.hi);
    X3_PHASE();
    six(acc[1][0], WB0, XA1);
    XA0 = split8(rXA0n.lo, rXA0n.hi);
    X3_PHASE();
    six(acc[1][1], WB1, XA1);
    WB0 = split8(rWB0n.lo, rWB0n.hi);
    X3_PHASE();
    stage = next_stage;
    if (++c_kt < nk) continue;

    // ---- epilogue of tile c_it
    int m0, n0;
    tile_origin(c_it, m0, n0);
    const bool vec_ok = (m0 + BM <= M) && (n0 + BN <= N) && (ldy & 3) == 0 && ((uintptr_t)Y & 15) == 0 && (!ADD || ((ldr & 3) == 0 && ((uintptr_t)R & 15) == 0));
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int m = m0 + wm * 64 + a * 32 + lr;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int n = n0 + wn * 64 + (g >> 2) * 32 + 8 * (g & 3) + 4 * lh;
        f32x4 val;
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = acc[a][g >> 2][4 * (g & 3) + j];
        if (vec_ok) {
          if (ADD) val += *reinterpret_cast<const f32x4*>(R + (int64_t)m * ldr + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (ACT == STLT_ACT_GELU) val[j] = gelu_bfree(val[j]);
            if (ACT == STLT_ACT_RELU) val[j] = fmaxf(val[j], 0.f);
          }
          *reinterpret_cast<f32x4*>(Y + (int64_t)m * ldy + n) = val;
        } else if (m < M) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (n + j < N) {
              float x = val[j];
              if (ADD) x += R[(int64_t)m * ldr + n + j];
              if (ACT == STLT_ACT_GELU) x = gelu_bfree(x);
              if (ACT == STLT_ACT_RELU) x = fmaxf(x, 0.f);
              Y[(int64_t)m * ldy + n + j] = x;
            }
          }
        }
      }
    }
    if (c_it + 1 < my_tiles) init_acc(c_it + 1);
    ++c_it;
    c_kt = 0;
  }
}

int g_split_bf16 = -1;  // -1: read STLT_GEMM_SPLIT_BF16 once; 0 off; 6 on

}  // namespace

extern "C" int stlt_set_gemm_split_bf16(int terms) {
  if (terms != 0 && terms != 6) return stlt_set_error(STLT_EINVAL, "stlt_set_gemm_split_bf16: 0 (off) or 6 (f32-equivalent six-term products)");
  g_split_bf16 = terms;
  return 0;
}

// *taken = true when the product was launched on the BF16 matrix cores; false: not enabled / not a shape of this kernel
int launch_linear_bf16x3(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* r, int64_t ldr, float* y,
                         int64_t ldy, int64_t M, int64_t N, int64_t K, int act, hipStream_t s, bool* taken) {
  *taken = false;
  if (g_split_bf16 < 0) { const char* e = getenv("STLT_GEMM_SPLIT_BF16"); g_split_bf16 = (e && atoi(e) == 6) ? 6 : 0; }
  if (g_split_bf16 != 6) return 0;
  if (K % BK != 0 || ldx % 4 != 0 || ldw % 4 != 0 || M <= 0 || N <= 0 || M > 0x7fffff00LL || N > 0x7fffff00LL) return 0;
  if (act != STLT_ACT_NONE && act != STLT_ACT_GELU && act != STLT_ACT_RELU) return 0;
  if (r && act != STLT_ACT_NONE) return 0;
  const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
  const int64_t cus = stlt_device_cus();
  if ((cus & 7) != 0 || n_tiles > 0x7fffffffLL) return 0;
  const int64_t rounds = (n_tiles + cus - 1) / cus;
  if ((double)n_tiles / (double)(rounds * cus) < 0.9) return 0;  // under-filled launches keep gemm.hip's stream-K
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)M * (double)N * (double)K);
  const dim3 grid((unsigned)cus), block(X_THREADS);
#define XL(ACTV, ADDV) hipLaunchKernelGGL((gemm_nt_bf16x3_kernel<ACTV, ADDV>), grid, block, 0, s, x, ldx, w, ldw, bias, r, ldr, y, ldy, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n)
  if (r) XL(STLT_ACT_NONE, true);
  else if (act == STLT_ACT_GELU) XL(STLT_ACT_GELU, false);
  else if (act == STLT_ACT_RELU) XL(STLT_ACT_RELU, false);
  else XL(STLT_ACT_NONE, false);
#undef XL
  *taken = true;
  return stlt_check_launch("gemm_nt_bf16x3_kernel");
}
