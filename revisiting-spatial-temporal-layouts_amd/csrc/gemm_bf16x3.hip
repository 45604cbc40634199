// nn.Linear forward  Y = act(X·Wᵀ + b) (+ R)  with f32 operands and an f32-equivalent result on the BF16 matrix cores
// ("split-bf16", opt-in: STLT_GEMM_SPLIT_BF16=6 / stlt_set_gemm_split_bf16; never the default and never bench.py's `value`).
//
// Every f32 operand element is cut into three bf16 pieces  a = a0 + a1 + a2  (a0 = the upper 16 bits of a, a1 = the upper 16
// bits of a - a0, a2 = a - a0 - a1 rounded to bf16: 24 significand bits in all, the subtractions are exact) and a product a·b
// is taken as the six piece products whose weight is at least 2^-24 of it,
//     a0·b0 + a0·b1 + a1·b0 + a1·b1 + a0·b2 + a2·b0,
// each a v_mfma_f32_16x16x32_bf16 accumulating in f32 (smallest terms first): products of bf16 pieces are exact in f32, the
// dropped terms (a1·b2, a2·b1, a2·b2) are below 2^-24 |a·b| — the rounding an f32 multiply makes anyway.  Measured error
// against an fp64 product: within eps_f32·sqrt(K) of the largest output on every shape tried (the bound of one sequential f32
// accumulation; 1 200 random shapes, profiles/round3_gemm_bf16x3_fuzz.txt); on the whole-tile forward shapes at bench sizes equal
// to the f32-MFMA kernel's within 15 % (profiles/round3_gemm_bf16x3.txt), on small launches up to ~9x the f32 kernel's, whose
// stream-K form sums K in short ranges.  Infinite operands give NaN where the f32 kernel gives +-inf (a - a0 of an infinity).
// The bf16 MFMA issues 16x the FLOPs of the f32 one per cycle, six of them make one f32-equivalent product: 2.7x on paper
// (419 TFLOP/s-equivalent).  Measured 215-235 = 1.62-1.79x the f32 kernel: the chip is power-bound under bf16 MFMAs (it runs
// this kernel at ~1.5 GHz) and after that the producer waves set the k-step; DESIGN.md section 3 has the numbers.
//
// Structure: 256 x 128 x 32 tiles, 8 MFMA waves (4 x 2 of 64 x 64 = 4 x 4 blocks of 16 x 16, transposed accumulators, bias as
// their initial value) and 4 PRODUCER waves, persistent workgroups walking gemm.hip's XCD-contiguous band-major tile order,
// one barrier per k-step.  The producers load the f32 operands with buffer loads into registers (two k-steps ahead, ~92 KB in
// flight per workgroup), cut every element ONCE per workgroup and write three bf16 plane images to LDS; the MFMA waves read
// ready bf16 fragments and issue nothing but ds_read_b128 and MFMAs.  Long launches cut the weight once for all workgroups in
// a kernel of their own (w_planes_kernel) and the producers only copy its planes.  History (git): every MFMA wave cutting its own fragments
// in registers (f750abb: 175-188 TFLOP/s, vector-issue bound); 32 x 32 x 16 blocks (180-199, power-bound: that MFMA form draws
// more per FLOP); a four-slot ring of 16-k steps (profiles/round3_gemm_bf16x3_ring_build.patch: no gain).
// Forward (NT) layout only, whole-tile launches only (no stream-K): shapes the launcher does not take keep gemm.hip's kernel.
#include <cstdlib>
#include "common.h"

#ifndef STLT_X3_STAMP
#define STLT_X3_STAMP 0
#endif
#ifndef STLT_X3_EXP
#define STLT_X3_EXP 0  // timing experiments (wrong results): 4 = producers write uncut bits, 8 = no operand loads after the prologue, 16 = one MFMA per block instead of six
#endif

namespace {

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int X_WAVES = 8, X_LOADERS = 4;
constexpr int X_THREADS = 64 * (X_WAVES + X_LOADERS);

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

struct Planes { u32x4 p0, p1, p2; };  // one operand fragment: eight bf16 per lane in each of the three piece planes
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// band-major, XCD-contiguous walk of the output tiles by persistent workgroups (gemm.hip's order): XCD x owns positions
// [x R Gx, (x+1) R Gx) and its workgroup l walks p = (x R + it) Gx + l
struct TileWalk {
  static constexpr int GROUP_M = 4;
  int tiles_m, tiles_n, g_xcd, g_local, g_gx, g_rounds, my_tiles;
  __device__ __forceinline__ TileWalk(int tm, int tn) : tiles_m(tm), tiles_n(tn) {
    const int n_tiles = tm * tn, G = gridDim.x;
    g_xcd = blockIdx.x & 7; g_local = blockIdx.x >> 3; g_gx = G >> 3;
    g_rounds = (n_tiles + G - 1) / G;
    my_tiles = 0;
    const int p0 = g_xcd * g_rounds * g_gx + g_local;
    if (p0 < n_tiles) my_tiles = (n_tiles - p0 + g_gx - 1) / g_gx;
    if (my_tiles > g_rounds) my_tiles = g_rounds;
  }
  __device__ __forceinline__ void origin(int it, int& m0, int& n0) const {
    const int p = (g_xcd * g_rounds + it) * g_gx + g_local;
    const int band = p / (GROUP_M * tiles_n), w = p - band * (GROUP_M * tiles_n);
    const int rows = tiles_m - band * GROUP_M < GROUP_M ? tiles_m - band * GROUP_M : GROUP_M;
    const int tn_g = w / rows;
    m0 = (band * GROUP_M + (w - tn_g * rows)) * BM;
    n0 = tn_g * BN;
  }
};

// the 64 x 64 block of MFMA wave (wm, wn) held as 4 x 4 blocks of 16 x 16 (v_mfma_f32_16x16x32_bf16, transposed: lane = output row lane & 15,
// its four registers = columns 4 (lane >> 4) .. + 3 of the block)
template <int ACT, bool ADD>
__device__ __forceinline__ void store_block16(const f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int lr16, int kq, const float* __restrict__ R,
                                              int64_t ldr, float* __restrict__ Y, int64_t ldy, int M, int N) {
  const bool vec_ok = (m0 + BM <= M) && (n0 + BN <= N) && (ldy & 3) == 0 && ((uintptr_t)Y & 15) == 0 && (!ADD || ((ldr & 3) == 0 && ((uintptr_t)R & 15) == 0));
#pragma unroll
  for (int bm = 0; bm < 4; ++bm) {
    const int m = m0 + wm * 64 + bm * 16 + lr16;
#pragma unroll
    for (int bn = 0; bn < 4; ++bn) {
      const int n = n0 + wn * 64 + bn * 16 + 4 * kq;
      f32x4 val = acc[bn][bm];
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
}

// LDS: two plane buffers of 72 KB (X planes 3 x 256 rows x 64 B, W planes 3 x 128 rows x 64 B; a row's four 16-byte groups
// at positions g ^ T[(row >> 2) & 3], T = {0, 3, 2, 1}: the 16-lane groups of ds_read_b128 — lanes {0-3, 12-15, 20-27}, ...
// (MI355X_MICROARCH.md LDS table) — then cover the 16 slots of the 256-byte bank row once each) + the bias strips.
constexpr int P_A_PLANE = BM * 64, P_B_PLANE = BN * 64;       // bytes of one plane image of one k-step
constexpr int P_B_BASE = 3 * P_A_PLANE;
constexpr int P_BUF = 3 * (P_A_PLANE + P_B_PLANE);            // 73728
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float x0 = v[2 * i], x1 = v[2 * i + 1];
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    p0[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    p1[i] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 t = {(__bf16)s0, (__bf16)s1};
    p2[i] = __builtin_bit_cast(unsigned, t);
  }
}

// WPRE: W arrives already cut — three bf16 planes [3][N][K] written by w_planes_kernel (the weight is shared by every row
// tile: cutting it once per launch takes a third of the cut off the producers) — in `W`, ldw unused
template <int ACT, bool ADD, bool WPRE>
__global__ __launch_bounds__(X_THREADS, 3) void gemm_nt_bf16x3p_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                                      int64_t ldw, const float* __restrict__ bias, const float* __restrict__ R,
                                                                      int64_t ldr, float* __restrict__ Y, int64_t ldy, int M, int N, int K,
                                                                      int tiles_m, int tiles_n, unsigned long long* __restrict__ dbg) {
  __shared__ __attribute__((aligned(16))) unsigned char pmem[2 * P_BUF + 2 * BN * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = K / BK;
  const TileWalk walk(tiles_m, tiles_n);
  const int my_tiles = walk.my_tiles;
  if (my_tiles <= 0) return;
  const int total_steps = my_tiles * nk;
  float* bias_lds = reinterpret_cast<float*>(pmem + 2 * P_BUF);
#if STLT_X3_STAMP  // diagnostic build: per-wave shader-clock totals -> dbg[(workgroup * 12 + wave) * 4 + {total, at the barrier, steps}]
  unsigned long long t_bar = 0, t_begin = __builtin_amdgcn_s_memtime();
  unsigned long long t_ph[6] = {0, 0, 0, 0, 0, 0}, t_last = t_begin;  // producer phases: [0] other (scalar, waits for operands), [1] cut, [2] plane stores, [3] load issue
#define X3_T(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_n = __builtin_amdgcn_s_memtime(); t_ph[k] += t_n - t_last; t_last = t_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define X3_BARRIER() do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_a = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); const unsigned long long t_b = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_bar += t_b - t_a; __builtin_amdgcn_sched_barrier(0); } while (0)
#define X3_FINISH() do { if (dbg && lane == 0) { unsigned long long* o = dbg + ((size_t)blockIdx.x * 12 + wave) * 4; o[0] = __builtin_amdgcn_s_memtime() - t_begin; o[1] = t_bar; o[2] = (unsigned long long)total_steps; if (wave >= X_WAVES) { unsigned long long* e = dbg + 256 * 12 * 4 + ((size_t)blockIdx.x * 4 + (wave - X_WAVES)) * 4; e[0] = t_ph[0]; e[1] = t_ph[1]; e[2] = t_ph[2]; e[3] = t_ph[3]; unsigned long long* f = dbg + 256 * 12 * 4 + 256 * 16 + ((size_t)blockIdx.x * 4 + (wave - X_WAVES)) * 2; f[0] = t_ph[4]; f[1] = t_ph[5]; } } } while (0)
#else
#define X3_T(k) do { } while (0)
#define X3_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define X3_FINISH() do { } while (0)
#endif

  if (wave >= X_WAVES) {
    // ---- producer waves: wave p owns X rows [64p, 64p+64) and W rows [32p, 32p+32); a lane holds 8 consecutive k of a row
    // (two 16-byte loads), so that each of its plane pieces is one 16-byte LDS store: 4 + 2 items of 16 rows per k-step
    const int p = wave - X_WAVES;
    const int prow = lane >> 2, pch = lane & 3;
    // operand loads = buffer loads: a per-tile descriptor in scalar registers (base = the tile's first row, extent = the rows
    // left in the matrix: rows past the edge read as zeros, no clamping), one lane-constant VGPR offset per operand, the
    // 8-row piece stride added per load and the k offset in the scalar offset field: no 64-bit lane addresses at all
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0, 0x27000);
    __amdgpu_buffer_rsrc_t rB = rA;
    auto extent = [](int64_t rows, int64_t ld) { const int64_t b = rows * ld * 4; return (int)(unsigned)(b > 0xffffffffLL ? 0xffffffffLL : b); };
    auto set_tile = [&](int it) {
      int m0, n0;
      walk.origin(it, m0, n0);
      rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (int64_t)m0 * ldx), 0, extent(M - m0, ldx), 0x27000);
      if (WPRE) {  // planes [3][N][K] of bf16: the tile's first row of plane 0; rows past N read the next plane or, past the end, zeros
        const int64_t off = (int64_t)n0 * K * 2, all = (int64_t)3 * N * K * 2;
        rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(W) + off), 0, (int)(unsigned)(all - off), 0x27000);
      } else {
        rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (int64_t)n0 * ldw), 0, extent(N - n0, ldw), 0x27000);
      }
    };
    const int ldx4 = (int)ldx * 4, ldw4 = (int)ldw * 4;
    const int vA = (p * 64 + prow) * ldx4 + pch * 32, vB = (p * 32 + prow) * ldw4 + pch * 32;
    const int vP = (p * 32 + prow) * (K * 2) + pch * 16, plane_bytes = N * K * 2;  // WPRE: a lane's 8 bf16 of a plane row
    constexpr int NPC = WPRE ? 14 : 12;  // register pieces per k-step: 8 of X + 4 of W (f32) or 6 of W (2 items x 3 planes, bf16)
    auto load_piece = [&](int j, int kt) {  // piece j < 8: half (j & 1) of X item j >> 1; then W
      if (j < 8) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, vA + (j >> 1) * 16 * ldx4 + (j & 1) * 16, kt * (BK * 4), 0));
      if (WPRE) {
        const int it = (j - 8) / 3, pl = (j - 8) % 3;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vP + it * 16 * (K * 2) + pl * plane_bytes, kt * (BK * 2), 0));
      }
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vB + ((j - 8) >> 1) * 16 * ldw4 + (j & 1) * 16, kt * (BK * 4), 0));
    };
    // bias strip: wave p carries columns [32p, 32p+32) of the NEXT tile's strip in one register, loaded every step and stored
    // (into the strip buffer the MFMA waves are not reading) one step later — no branch and no wait of its own in this wave's
    // load stream, so the counted waits the compiler derives for the operand registers stay exact.  The store of a tile's first
    // step still carries the previous strip; the later steps overwrite it (K >= 2 k-steps, the launcher's condition)
    auto bias_col = [&](int it) {  // this lane's column of tile it's strip (clamped to the matrix); two scalar divisions: once per tile
      it = it < my_tiles ? it : my_tiles - 1;
      int m0, n0;
      walk.origin(it, m0, n0);
      const int n = n0 + 32 * p + (lane & 31);
      return n < N ? n : N - 1;
    };
    auto bias_load = [&](int n) { return bias ? bias[n] : 0.f; };
    auto bias_store = [&](int it, float v) { bias_lds[(it & 1) * BN + 32 * p + (lane & 31)] = v; };
    int l_it = 0, l_kt = 0;
    auto load_step = [&](f32x4 (&S)[NPC]) {
      if (l_kt == 0 && l_it < my_tiles) set_tile(l_it);
#pragma unroll
      for (int j = 0; j < NPC; ++j) S[j] = load_piece(j, l_kt);
      if (++l_kt == nk) { ++l_it; l_kt = 0; }
    };
    // cut the step held in S into plane buffer `buf`, refilling every 16-byte register group with the same piece of the step
    // two k-steps later the moment it has been cut: ~92 KB of operands in flight per workgroup on 96 registers per lane
    auto cut_step = [&](f32x4 (&S)[NPC], int buf, bool reload) {
      unsigned char* base = pmem + buf * P_BUF + prow * 64;
      if (reload && l_kt == 0 && l_it < my_tiles) set_tile(l_it);  // past the last step: the last tile's k-steps again (never read)
      const int grp = (pch ^ ((0 - (prow >> 2)) & 3)) * 16;  // the fragment layout's swizzle: group ^ T[(row >> 2) & 3], T = {0, 3, 2, 1}
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        unsigned char* d = (q < 4 ? base + (p * 64 + q * 16) * 64 : base + P_B_BASE + (p * 32 + (q - 4) * 16) * 64) + grp;
        const int ps = q < 4 ? P_A_PLANE : P_B_PLANE;
        if (WPRE && q >= 4) {  // ready planes: straight to LDS
          const int j0 = 8 + (q - 4) * 3;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(d + pl * ps) = __builtin_bit_cast(u32x4, S[j0 + pl]);
          X3_T(2);
#if !(STLT_X3_EXP & 8)
          if (reload) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) S[j0 + pl] = load_piece(j0 + pl, l_kt);
          }
#endif
          X3_T(3);
          continue;
        }
        u32x2 a0, a1, a2, b0, b1, b2;
#if STLT_X3_EXP & 4  // timing experiment: producers write uncut bits
        a0 = u32x2{__builtin_bit_cast(unsigned, S[2 * q][0]), __builtin_bit_cast(unsigned, S[2 * q][1])}; a1 = u32x2{__builtin_bit_cast(unsigned, S[2 * q][2]), __builtin_bit_cast(unsigned, S[2 * q][3])}; a2 = a0;
        b0 = u32x2{__builtin_bit_cast(unsigned, S[2 * q + 1][0]), __builtin_bit_cast(unsigned, S[2 * q + 1][1])}; b1 = u32x2{__builtin_bit_cast(unsigned, S[2 * q + 1][2]), __builtin_bit_cast(unsigned, S[2 * q + 1][3])}; b2 = b0;
#else
        X3_T(0);
        split4(S[2 * q], a0, a1, a2);
        split4(S[2 * q + 1], b0, b1, b2);
        X3_T(1);
#endif
        *reinterpret_cast<u32x4*>(d) = u32x4{a0[0], a0[1], b0[0], b0[1]};
        *reinterpret_cast<u32x4*>(d + ps) = u32x4{a1[0], a1[1], b1[0], b1[1]};
        *reinterpret_cast<u32x4*>(d + 2 * ps) = u32x4{a2[0], a2[1], b2[0], b2[1]};
        X3_T(2);
#if !(STLT_X3_EXP & 8)  // timing experiment: no operand loads after the first two steps
        if (reload) { S[2 * q] = load_piece(2 * q, l_kt); S[2 * q + 1] = load_piece(2 * q + 1, l_kt); }
#endif
        X3_T(3);
      }
      if (reload) { if (++l_kt == nk) { ++l_it; l_kt = 0; } }
    };
    f32x4 S0[NPC], S1[NPC];
    bias_store(0, bias_load(bias_col(0)));
    int n_next = bias_col(1);
    float b_next = bias_load(n_next);
    load_step(S0);
    load_step(S1);
    cut_step(S0, 0, true);  // step 0 -> buffer 0; S0 <- step 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int w_it = 0, w_kt = 0;
    auto iter = [&](int i, f32x4 (&S)[NPC]) {  // while the MFMA waves multiply step i: cut step i + 1, load step i + 3
      X3_T(0);
      bias_store(w_it + 1, b_next);
      X3_T(4);
      b_next = bias_load(n_next);
      X3_T(5);
      cut_step(S, (i + 1) & 1, true);
      X3_BARRIER();
      if (++w_kt == nk) { ++w_it; w_kt = 0; n_next = bias_col(w_it + 1); }
    };
    for (int i = 0; i < total_steps; i += 2) {
      iter(i, S1);
      if (i + 1 >= total_steps) break;
      iter(i + 1, S0);
    }
    X3_FINISH();
    return;
  }

  // ---- MFMA waves
  const int wm = wave >> 1, wn = wave & 1;
  // 16 blocks of 16 x 16 per wave; lane (r = lane & 15, kq = lane >> 4) holds k = 8 kq .. + 7 of row r of a fragment: one
  // ds_read_b128 per plane, one MFMA per piece pair and k-step.  The row's four 16-byte groups sit at kq ^ T[(row >> 2) & 3],
  // T = {0, 3, 2, 1}: every 16-lane group of ds_read_b128 ({0-3, 12-15, 20-27}, ...) then covers the bank row's 16 slots once.
  // (In bare MFMA loops the 16x16x32 form delivers ~1.15x the FLOP/s of 32x32x16 at the same cycles per FLOP — it draws less
  // power, MI355X_MICROARCH.md — and this kernel is power-bound.)
  const int lr16 = lane & 15, kq = lane >> 4;
  const int sw16 = (kq ^ ((0 - (lr16 >> 2)) & 3)) * 16;
  const int a_off = (wm * 64 + lr16) * 64 + sw16, b_off = P_B_BASE + (wn * 64 + lr16) * 64 + sw16;
  f32x4 acc[4][4];  // [bn][bm]
  auto init_acc = [&](int it) {
#pragma unroll
    for (int bn = 0; bn < 4; ++bn) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (bias) v = *reinterpret_cast<const f32x4*>(bias_lds + (it & 1) * BN + wn * 64 + bn * 16 + 4 * kq);
#pragma unroll
      for (int bm = 0; bm < 4; ++bm) acc[bn][bm] = v;
    }
  };
  auto read_planes = [&](int buf, int off, int ps) {
    const unsigned char* s = pmem + buf * P_BUF + off;
    Planes f;
    f.p0 = *reinterpret_cast<const u32x4*>(s);
    f.p1 = *reinterpret_cast<const u32x4*>(s + ps);
    f.p2 = *reinterpret_cast<const u32x4*>(s + 2 * ps);
    return f;
  };
  auto six = [&](f32x4& d, const Planes& w, const Planes& x) {
#if !(STLT_X3_EXP & 16)
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w.p2), as_bf(x.p0), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w.p0), as_bf(x.p2), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w.p1), as_bf(x.p1), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w.p1), as_bf(x.p0), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w.p0), as_bf(x.p1), d, 0, 0, 0);
#endif
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(w.p0), as_bf(x.p0), d, 0, 0, 0);
  };

  __builtin_amdgcn_s_barrier();
  init_acc(0);
  int c_it = 0, c_kt = 0;
  for (int step = 0; step < total_steps; ++step) {
    const int buf = step & 1;
    Planes wf[4];
#pragma unroll
    for (int bn = 0; bn < 4; ++bn) wf[bn] = read_planes(buf, b_off + bn * 16 * 64, P_B_PLANE);
#pragma unroll
    for (int bm = 0; bm < 4; ++bm) {
      const Planes xf = read_planes(buf, a_off + bm * 16 * 64, P_A_PLANE);
#pragma unroll
      for (int bn = 0; bn < 4; ++bn) six(acc[bn][bm], wf[bn], xf);
    }
    X3_BARRIER();
    if (++c_kt < nk) continue;
    int m0, n0;
    walk.origin(c_it, m0, n0);
    store_block16<ACT, ADD>(acc, m0, n0, wm, wn, lr16, kq, R, ldr, Y, ldy, M, N);
    if (c_it + 1 < my_tiles) init_acc(c_it + 1);
    ++c_it;
    c_kt = 0;
  }
  X3_FINISH();
}

// W (n_out, k_in) -> Wt (k_in, n_out): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, int n_out, int k_in, float* __restrict__ wt) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, k = k0 + tx;
    if (n < n_out && k < k_in) tile[ty + 8 * i][tx] = w[(int64_t)n * k_in + k];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, n = n0 + tx;
    if (n < n_out && k < k_in) wt[(int64_t)k * n_out + n] = tile[tx][ty + 8 * i];
  }
}


// W (N, K) f32 -> three bf16 planes [3][N][K] (the pieces split4 makes), once per launch: 4 k per thread
__global__ __launch_bounds__(256) void w_planes_kernel(const float* __restrict__ w, int64_t ldw, int N, int K, unsigned char* __restrict__ planes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (row, group of 4 k)
  const int kg = K >> 2;
  if (i >= (int64_t)N * kg) return;
  const int n = (int)(i / kg), k = (int)(i - (int64_t)n * kg) * 4;
  u32x2 p0, p1, p2;
  split4(*reinterpret_cast<const f32x4*>(w + (int64_t)n * ldw + k), p0, p1, p2);
  const int64_t off = ((int64_t)n * K + k) * 2, pb = (int64_t)N * K * 2;
  *reinterpret_cast<u32x2*>(planes + off) = p0;
  *reinterpret_cast<u32x2*>(planes + pb + off) = p1;
  *reinterpret_cast<u32x2*>(planes + 2 * pb + off) = p2;
}

int g_split_bf16 = -1;  // -1: read STLT_GEMM_SPLIT_BF16 once; 0 off; 6 on

}  // namespace

extern "C" int stlt_set_gemm_split_bf16(int terms) {
  if (terms != 0 && terms != 6) return stlt_set_error(STLT_EINVAL, "stlt_set_gemm_split_bf16: 0 (off) or 6 (f32-equivalent six-term products)");
  g_split_bf16 = terms;
  return 0;
}

// *taken = true when the product was launched on the BF16 matrix cores; false: not enabled / not a shape of this kernel
// does the split-bf16 kernel take this nn.Linear forward (when it is switched on)?  Whole-tile launches of a contraction of at
// least two k-steps whose rounds of whole tiles fill at least half of the chip's workgroups (STLT_X3_MIN_FILL, default 0.5:
// measured on the cfg2 forward at 16 / 64 / 256 clips and cfg4 at 64, thresholds 0.3 - 0.6 are within 1 % of each other, 0.9
// and 0.15 lose 10 - 20 %); below that gemm.hip's stream-K kernel keeps the launch
bool stlt_split_bf16_takes(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw) {
  if (g_split_bf16 < 0) { const char* e = getenv("STLT_GEMM_SPLIT_BF16"); g_split_bf16 = (e && atoi(e) == 6) ? 6 : 0; }
  if (g_split_bf16 != 6) return false;
  if (K % BK != 0 || K < 2 * BK || ldx % 4 != 0 || ldw % 4 != 0 || ldx >= (1 << 21) || ldw >= (1 << 21) || M <= 0 || N <= 0 || M > 0x7fffff00LL || N > 0x7fffff00LL) return false;
  const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
  const int64_t cus = stlt_device_cus();
  if ((cus & 7) != 0 || n_tiles > 0x7fffffffLL) return false;
  const int64_t rounds = (n_tiles + cus - 1) / cus;
  static const double min_fill = [] { const char* e = getenv("STLT_X3_MIN_FILL"); return e ? atof(e) : 0.5; }();
  return (double)n_tiles / (double)(rounds * cus) >= min_fill;
}

int launch_linear_bf16x3(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* r, int64_t ldr, float* y,
                         int64_t ldy, int64_t M, int64_t N, int64_t K, int act, hipStream_t s, bool* taken) {
  *taken = false;
  if (!stlt_split_bf16_takes(M, N, K, ldx, ldw)) return 0;
  if (act != STLT_ACT_NONE && act != STLT_ACT_GELU && act != STLT_ACT_RELU) return 0;
  if (r && act != STLT_ACT_NONE) return 0;
  // the argument checks launch_gemm makes for the f32 kernel: with the opt-in switch on, a bad call must still come back as
  // STLT_EINVAL, not as an out-of-bounds device access (pointers, row pitches below the row length)
  if (!x || !w || !y) return stlt_set_error(STLT_EINVAL, "gemm (split-bf16): null pointer");
  if (ldx < K || ldw < K || ldy < N || (r && ldr < N))
    return stlt_set_error(STLT_EINVAL, "gemm (split-bf16): bad leading dimension (ldx=%lld ldw=%lld ldy=%lld ldr=%lld for N=%lld K=%lld)", (long long)ldx,
                          (long long)ldw, (long long)ldy, (long long)ldr, (long long)N, (long long)K);
  const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int64_t cus = stlt_device_cus();
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)M * (double)N * (double)K);
  const dim3 grid((unsigned)cus), block(X_THREADS);
  // the weight cut once per launch into the lent stream-K scratch (free between launches, stream-ordered) when the launch is
  // long enough to pay for the extra 5-10 us kernel: at least four rounds of tiles (STLT_X3_WPRE=0 keeps the producers' own cut).
  // Measured on the forward shapes at 1024 clips: 227 - 242 TFLOP/s against 221 - 235 (+2.5 %)
  static const bool wpre_on = [] { const char* e = getenv("STLT_X3_WPRE"); return e ? atoi(e) != 0 : true; }();
  size_t sbytes = 0;
  unsigned char* planes = reinterpret_cast<unsigned char*>(stlt_gemm_scratch_ptr(&sbytes));
  const bool wpre = wpre_on && planes && (size_t)3 * (size_t)N * (size_t)K * 2 <= sbytes && tiles_m * tiles_n >= 4 * cus &&
                    (int64_t)3 * N * K * 2 < 0x7fffffffLL;
  if (wpre) {
    const int64_t n_thr = N * (K / 4);
    hipLaunchKernelGGL(w_planes_kernel, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, s, w, ldw, (int)N, (int)K, planes);
    if (int e = stlt_check_launch("w_planes_kernel")) return e;
  }
  const float* wsrc = wpre ? reinterpret_cast<const float*>(planes) : w;
#define XL(ACTV, ADDV) do { if (wpre) hipLaunchKernelGGL((gemm_nt_bf16x3p_kernel<ACTV, ADDV, true>), grid, block, 0, s, x, ldx, wsrc, ldw, bias, r, ldr, y, ldy, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n, g_stlt_debug_buf); \
  else hipLaunchKernelGGL((gemm_nt_bf16x3p_kernel<ACTV, ADDV, false>), grid, block, 0, s, x, ldx, wsrc, ldw, bias, r, ldr, y, ldy, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n, g_stlt_debug_buf); } while (0)
  if (r) XL(STLT_ACT_NONE, true);
  else if (act == STLT_ACT_GELU) XL(STLT_ACT_GELU, false);
  else if (act == STLT_ACT_RELU) XL(STLT_ACT_RELU, false);
  else XL(STLT_ACT_NONE, false);
#undef XL
  *taken = true;
  return stlt_check_launch("gemm_nt_bf16x3p_kernel");
}

// input gradient of a Linear, C (rows, k_in) = dY (rows, n_out)·W (n_out, k_in) (+ R), on the split-bf16 kernel: W is transposed
// into wt_scratch (n_out * k_in floats; 5-10 us for 0.6-2.4 M elements) and the product runs as the NT form above.
// *taken = false (nothing launched) when the switch is off or the launch does not qualify.
int launch_input_grad_bf16x3(const float* dy, int64_t ld_dy, const float* w, int64_t n_out, int64_t k_in, const float* r, int64_t ldr, float* c,
                             int64_t ldc, int64_t rows, float* wt_scratch, hipStream_t s, bool* taken) {
  *taken = false;
  if (!wt_scratch || !stlt_split_bf16_takes(rows, k_in, n_out, ld_dy, n_out)) return 0;
  if (!dy || !w || !c) return stlt_set_error(STLT_EINVAL, "input gradient (split-bf16): null pointer");
  if (ld_dy < n_out || ldc < k_in || (r && ldr < k_in)) return stlt_set_error(STLT_EINVAL, "input gradient (split-bf16): bad leading dimension");
  hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)((k_in + 31) / 32), (unsigned)((n_out + 31) / 32)), dim3(256), 0, s, w, (int)n_out, (int)k_in, wt_scratch);
  if (int e = stlt_check_launch("weight_transpose_kernel")) return e;
  return launch_linear_bf16x3(dy, ld_dy, wt_scratch, n_out, nullptr, r, ldr, c, ldc, rows, k_in, n_out, STLT_ACT_NONE, s, taken);
}
