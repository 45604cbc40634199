// The small-tile product kernel (see gemm16.hip for the why and the routing): nn.Linear forward  Y = act(X·Wᵀ + b) (+ R)  and, WKN build,
// the input gradient dX = dY·W, on WHOLE tiles of (16 RB) rows x (16 NT) columns, RB in {2, 4, 8}.  Shared by gemm16.hip (128-row tiles),
// gemm16_rb4.hip (64 rows) and gemm16_rb2.hip (32 rows): one translation unit per tile height so that the ~100 instantiations compile in
// parallel.
//
// Structure (mhsa.hip's product phase): v_mfma_f32_16x16x4_f32; 8 MFMA waves arranged RB x CG (CG = 8 / RB): wave w owns the 16-row block
// w % RB and the NTW = NT / CG column tiles of column group w / RB, transposed accumulators (lane = row, registers = 4 consecutive
// columns -> 16-byte stores); 4 DMA-only loader waves two to five k-steps ahead (LDS-DMA with the source-side bank swizzle, as many
// stages as fit ~150 KB, counted vmcnt, one barrier per k-step); bias as the accumulators' initial value from LDS strips DMA'd in front
// of each tile's first k-step; persistent workgroups over XCD-contiguous tile ranges (column tile fastest, so that the workgroups of an
// XCD share X row panels).  K % 32 == 0, N % 4 == 0.
#pragma once
#include "common.h"

#ifndef STLT_G16_SCHED
#define STLT_G16_SCHED 0  // 1: the k-step's read / MFMA phases pinned with sched_barrier; 0: the compiler's own order (measured equal on narrow tiles, 3 - 8 % faster on 192-column ones: profiles/round5_gemm16_ablation.txt)
#endif
#ifndef STLT_G16_DEEP
#define STLT_G16_DEEP 1  // 1: narrow wave tiles prefetch a whole k-step of fragments (0: the two-phase pipeline for every tile, A/B builds)
#endif
#ifndef STLT_G16_PRIO
#define STLT_G16_PRIO 0  // s_setprio of the MFMA waves (the loader waves stay at 0); A/B builds
#endif
#ifndef STLT_G16_ABLATE
#define STLT_G16_ABLATE 0  // timing-only builds (wrong results): bit 0 no steady-state DMA, bit 1 no steady-state fragment reads, bit 2 no steady-state barrier, bit 3 no epilogue stores, bit 4 the loaders re-read k-step 0 (cache-hot source), bit 5 the loaders run their code without the DMA instructions
#endif

namespace g16 {

constexpr int QK = 32;
constexpr int Q_WAVES = 8, Q_LOADERS = 4;
constexpr int Q_THREADS = 64 * (Q_WAVES + Q_LOADERS);
constexpr int Q_BIAS_STRIPS = 4;  // see dma_bias

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

struct Gemm16Args {
  const float* X; const float* W; const float* bias; const float* R;
  float* Y;
  int64_t ldx, ldw, ldr, ldy;
  int M, N, K, tiles_m, tiles_n;
  StltGemmEpi epi;  // ACT == STLT_ACT_GELU_BWD (R = the pre-activation u, read, not added) / STLT_ACT_GELU_KEEP (R = where u is stored; dr, site, drop_rows)
};

template <int RB, int NT> constexpr int q_stage_floats() { return (16 * RB + 16 * NT) * QK; }
// operand stages: as many as fit ~150 KB, at most 6.  A narrow tile's k-step is short (128 x 48: ~0.7 us), so the loaders must run more
// than two k-steps ahead to cover a miss in the L2 (measured with three stages: ~7 us lost per tile)
template <int RB, int NT> constexpr int q_nstage() {
  return 150 * 1024 / (q_stage_floats<RB, NT>() * 4) > 6 ? 6 : 150 * 1024 / (q_stage_floats<RB, NT>() * 4);
}
template <int RB, int NT> constexpr int q_smem_floats() { return q_nstage<RB, NT>() * q_stage_floats<RB, NT>() + Q_BIAS_STRIPS * 16 * NT; }

// WKN = false: W is (N, K) row-major (nn.Linear's weight; forward products).  WKN = true: W is (K, N) row-major — the input-gradient
// product dX = dY·W of a Linear whose weight (n_out, k_in) is read as it lies, contraction over its rows: the W image in LDS is then
// [32 k][BN n], both operands' fragments are gathered with the k order (16 c + lg + 4 e) so that a lane's four values of a column tile
// are four rows of that image (ds_read_b32; bank-conflict-free for 48 / 144 columns, 2-way otherwise) — no transposed weight copy.
template <int RB, int NT, int ACT, bool ADD, bool WKN>
__global__ __launch_bounds__(Q_THREADS, 3) void gemm16_kernel(const Gemm16Args a) {
  static_assert(RB == 2 || RB == 4 || RB == 8, "tile height: 32, 64 or 128 rows");
  constexpr int CG = Q_WAVES / RB;  // column groups of the MFMA waves
  static_assert(NT % CG == 0, "every column group owns the same number of column tiles");
  constexpr int NTW = NT / CG;      // column tiles per wave
  constexpr int QM = 16 * RB, BN = 16 * NT;
  constexpr int STAGE = q_stage_floats<RB, NT>();
  constexpr int Q_NSTAGE = q_nstage<RB, NT>();
  static_assert(Q_NSTAGE >= 3, "at least three operand stages");
  constexpr int LA = Q_NSTAGE - 1;  // k-steps the loaders run ahead of the MFMA waves
  // Narrow wave tiles (<= 3 column tiles per wave: 12 MFMAs per k-chunk) cannot cover an LDS round trip with the MFMAs of one phase.  Their
  // MFMA waves keep a whole k-step of fragments in registers and request step s + 1's while multiplying step s, so the barrier that ends
  // step s must already have published step s + 2: the loaders' counted wait leaves one step fewer in flight (DEEP; needs >= 4 stages).
  constexpr bool DEEP = STLT_G16_DEEP && NTW <= 3 && Q_NSTAGE >= 4;
  constexpr int PUB = DEEP ? 2 : 1;  // the barrier at the end of step s publishes step s + PUB
  constexpr int NI = 2 * RB + 2 * NT;   // 8-row (1 KB) LDS-DMA instructions per k-step: the X image's 2 RB, then the W image's 2 NT
  constexpr int NL_MAX = (NI + 3) / 4;  // ... dealt round-robin to the four loaders
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = a.K / QK;
  const int n_tiles = a.tiles_m * a.tiles_n;
  const int G = gridDim.x;
  // contiguous tile range per workgroup, workgroups in XCD-contiguous order (round-robin dispatch: blockIdx & 7 = XCD)
  int v = blockIdx.x;
  if ((G & 7) == 0) v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int per = n_tiles / G, extra = n_tiles - per * G;
  const int tile0 = v * per + (v < extra ? v : extra);
  const int my_items = per + (v < extra ? 1 : 0);
  if (my_items <= 0) return;
  const int total_steps = my_items * nk;
  float* bias_lds = smem + Q_NSTAGE * STAGE;
  auto item_of = [&](int it, int& tm, int& tn) {
    const int t = tile0 + it;
    tm = t / a.tiles_n;
    tn = t - tm * a.tiles_n;
  };

  if (wave >= Q_WAVES) {
    // ---- loader waves.  Instruction j of a k-step fills floats [256 j, 256 j + 256) of the stage (the W image follows the X image):
    // j < 2 RB: rows 8 j .. 8 j + 7 of the X image; else rows 8 (j - 2 RB) .. of the W image.  Loader Ld issues j = Ld, Ld + 4, ...
    const int Ld = wave - Q_WAVES;
    const int drow = lane >> 3, dslot = lane & 7;
    const int n_mine = (NI - Ld + 3) / 4;  // this loader's instructions per k-step (NL_MAX or NL_MAX - 1)
    // Addresses are a wave-uniform base per operand (the tile's first row at the current k-step: scalar registers, advanced by scalar
    // adds) plus a per-lane 32-bit byte offset fixed for the tile: the steady state issues no vector-ALU instruction at all.  (Round 4's
    // loaders kept 64-bit per-lane pointers and added the k offset per instruction; the loader code alone — DMA instructions taken out —
    // cost the MFMA waves of its SIMD 4 - 6 % of a narrow tile's k-step: profiles/round5_gemm16_ablation.txt.)
    uint32_t vo[NL_MAX];
    const char* bx = nullptr;
    const char* bw = nullptr;
    auto set_item = [&](int it) {
      int tm, tn;
      item_of(it, tm, tn);
      const int row0 = tm * QM, col0 = tn * BN;
      bx = reinterpret_cast<const char*>(a.X + (int64_t)row0 * a.ldx);
      bw = reinterpret_cast<const char*>(WKN ? a.W + col0 : a.W + (int64_t)col0 * a.ldw);
      const uint32_t ldx = (uint32_t)a.ldx, ldw = (uint32_t)a.ldw;
#pragma unroll
      for (int i = 0; i < NL_MAX; ++i) {
        const int j = Ld + 4 * i;
        if (j < 2 * RB) {
          const int r = j * 8 + drow;
          const int rr = r < a.M - row0 ? r : a.M - 1 - row0;  // rows past the matrix re-read the last row; their outputs are never stored
          vo[i] = ((uint32_t)rr * ldx + (uint32_t)((dslot ^ ((r >> 1) & 7)) * 4)) * 4u;
        } else if (WKN) {  // floats [256 jj, 256 jj + 256) of the [32 k][BN n] image: lane -> (k, n .. n + 3)
          const int f = ((j - 2 * RB) * 64 + lane) * 4;
          const int k = f / BN, n = f - k * BN;
          const int nn = col0 + n + 4 <= a.N ? n : a.N - 4 - col0;  // columns past the matrix re-read its last four; their outputs are never stored
          vo[i] = ((uint32_t)k * ldw + (uint32_t)nn) * 4u;
        } else {
          const int r = (j - 2 * RB) * 8 + drow;
          const int rr = r < a.N - col0 ? r : a.N - 1 - col0;
          vo[i] = ((uint32_t)rr * ldw + (uint32_t)((dslot ^ ((r >> 1) & 7)) * 4)) * 4u;
        }
      }
    };
    // Bias strip of tile `it` (its accumulators' initial value), issued IN FRONT of the tile's first k-step: the counter is in order, so
    // the wait that publishes that k-step publishes the strip as well.  (Until the end of round 4 the strip was issued one k-step before
    // it was read, behind up to LA - 1 newer steps the counted wait lets stay in flight: a race that was almost always won — the 3 KB of
    // bias are L2-resident — and lost once in a 33 000-row launch of a test run.)  Four strips: the loaders are LA <= 5 steps ahead and a
    // tile has nk >= 2 k-steps, so a strip is rewritten at the earliest 4 nk - LA >= 3 barriers after the MFMA waves read it.
    auto dma_bias = [&](int it) {  // BN bias values, 64 per instruction, dealt over the loaders
      if (a.bias && Ld * 64 < BN) {
        int tm, tn;
        item_of(it, tm, tn);
        int gn = tn * BN + Ld * 64 + lane;
        gn = gn < a.N ? gn : a.N - 1;  // columns past the matrix re-read its last one (N < 2^30: the byte offset fits 32 bits)
        if (Ld * 64 + lane < BN) stlt_dma4(a.bias, (uint32_t)gn * 4u, stlt_lds_addr(bias_lds + (it & (Q_BIAS_STRIPS - 1)) * BN + Ld * 64));
      }
    };
    static_assert(BN <= 256, "the bias strip is dealt as one 64-column instruction per loader");
    int l_it = 0, l_kt = 0, l_stage = 0;
    auto l_step = [&]() {
      if (l_kt == 0) { set_item(l_it); dma_bias(l_it); }
      float* st = smem + l_stage * STAGE;
#pragma unroll
      for (int i = 0; i < NL_MAX; ++i) {
        const int j = Ld + 4 * i;
        if (i < n_mine) {
          if (STLT_G16_ABLATE & 32) asm volatile("" :: "v"(vo[i]), "s"(j < 2 * RB ? bx : bw), "s"(st));
          else stlt_dma16(j < 2 * RB ? bx : bw, vo[i], stlt_lds_addr(st + j * 256));
        }
      }
      if (!(STLT_G16_ABLATE & 16)) {
        bx += QK * sizeof(float);
        bw += WKN ? (int64_t)QK * a.ldw * (int64_t)sizeof(float) : (int64_t)(QK * sizeof(float));
      }
      if (++l_kt == nk) { ++l_it; l_kt = 0; }
      if (++l_stage == Q_NSTAGE) l_stage = 0;
    };
    // in-order counter: once at most the instructions of the newest LA - PUB steps are in flight, everything up to the step the MFMA
    // waves read next has landed, and the bias strip in front of it (a strip among the newer instructions only makes the wait stricter).
    // The count per step is a per-loader constant, so the wait is one of two immediates.
    constexpr int WAIT_FULL = (LA - PUB) * NL_MAX, WAIT_LESS = (LA - PUB) * (NL_MAX - 1);
    static_assert(WAIT_FULL < 64, "vmcnt is a 6-bit counter");
    auto wait_ahead = [&]() {
      if (n_mine == NL_MAX) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAIT_FULL) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAIT_LESS) : "memory");
    };
    if (total_steps >= LA) {
#pragma unroll
      for (int i = 0; i < LA; ++i) l_step();
      wait_ahead();
    } else {
      for (int i = 0; i < total_steps; ++i) l_step();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    for (int step = 0; step < total_steps; ++step) {
      if (step + LA < total_steps) {
        if (!(STLT_G16_ABLATE & 1)) l_step();
        wait_ahead();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (!(STLT_G16_ABLATE & 4)) __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ---- MFMA waves: wave w owns rows [16 rb, 16 rb + 16) of the tile and the column tiles cg NTW .. cg NTW + NTW - 1
  if (STLT_G16_PRIO) __builtin_amdgcn_s_setprio(STLT_G16_PRIO);
  const int rb = wave % RB, cg = wave / RB;
  const int li = lane & 15, lg = lane >> 4;
  const int sw = (li >> 1) & 7;
  const int x_row = (rb * 16 + li) * QK;
  const int w_row = (QM + cg * NTW * 16 + li) * QK;
  // A k-step is NP phases: one k-chunk of 16 per phase with all the wave's column tiles (NTW <= 6), or half the column tiles per
  // phase (registers).  The fragments of phase p + 1 are requested behind the first MFMA group of phase p, and those of the next stage's
  // phase 0 right behind the barrier that publishes it, in front of the last phase's MFMAs.  (STLT_G16_SCHED=1 pins that order with
  // sched_barrier; the compiler's own placement measured equal on narrow tiles and 3 - 8 % faster on 192-column ones.)
  constexpr bool SPLIT = NTW > 6;
  constexpr int NP = SPLIT ? 4 : 2;
  constexpr int PT = SPLIT ? (NTW + 1) / 2 : NTW;  // column tiles per phase (the second half of a split chunk has NTW - PT)
  struct Frags { f32x4 w[PT]; };
  f32x4 xf[2];  // X fragment of k-chunk c: read once per chunk, shared by the chunk's two halves
  Frags F[2];
  auto read_phase = [&](int stage, int p, Frags& f) {
    const float* s = smem + stage * STAGE;
    const int c = SPLIT ? p >> 1 : p;
    const int t0 = SPLIT ? PT * (p & 1) : 0;
    const bool first_of_chunk = !SPLIT || (p & 1) == 0;
    if (WKN) {  // k order 16 c + lg + 4 e for MFMA e: element lg of X chunk 4 c + e; row 16 c + lg + 4 e of the [k][n] W image
      if (first_of_chunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xf[c][e] = s[x_row + (((4 * c + e) ^ sw) * 4) + lg];
      }
      const float* wk = s + QM * QK + (16 * c + lg) * BN + cg * NTW * 16 + li;
#pragma unroll
      for (int t = 0; t < PT; ++t)
        if (t0 + t < NTW) {
#pragma unroll
          for (int e = 0; e < 4; ++e) f.w[t][e] = wk[4 * e * BN + (t0 + t) * 16];
        }
      return;
    }
    const int off = ((4 * c + lg) ^ sw) * 4;
    if (first_of_chunk) xf[c] = *reinterpret_cast<const f32x4*>(s + x_row + off);
#pragma unroll
    for (int t = 0; t < PT; ++t)
      if (t0 + t < NTW) f.w[t] = *reinterpret_cast<const f32x4*>(s + w_row + (t0 + t) * 16 * QK + off);
  };
  f32x4 acc[NTW];
  auto init_acc = [&](int it) {
    if (a.bias) {
      const float* src = bias_lds + (it & (Q_BIAS_STRIPS - 1)) * BN + cg * NTW * 16 + 4 * lg;
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[t] = *reinterpret_cast<const f32x4*>(src + 16 * t);
    } else {
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto mfma_phase = [&](int p, const Frags& f, int e_lo, int e_hi) {  // e outermost: an accumulator recurs every (tiles in the phase) MFMAs
    const int c = SPLIT ? p >> 1 : p;
    const int t0 = SPLIT ? PT * (p & 1) : 0;
#pragma unroll
    for (int e = e_lo; e < e_hi; ++e)
#pragma unroll
      for (int t = 0; t < PT; ++t)
        if (t0 + t < NTW) acc[t0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[t][e], xf[c][e], acc[t0 + t], 0, 0, 0);
  };
#if STLT_G16_SCHED
#define G16_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define G16_PIN() ((void)0)
#endif

  int c_it = 0, c_kt = 0, stage = 0;
  // epilogue of tile c_it (its last k-step's MFMAs are issued), then the next tile's accumulators
  auto tile_done = [&](int step) {
    // ---- epilogue of tile c_it: lane (li, lg) holds columns 16 (cg NTW + t) + 4 lg .. + 3 of row 16 rb + li
    int tm, tn;
    item_of(c_it, tm, tn);
    {
      int eli = li, elg = lg;  // opaque copies: the address arithmetic is recomputed per tile instead of living in registers across the k-loop
      asm volatile("" : "+v"(eli), "+v"(elg));
      const int row = tm * QM + rb * 16 + eli;
      const int col0 = tn * BN + cg * NTW * 16 + 4 * elg;
      if constexpr (ACT == STLT_ACT_GELU_BWD) {
        // the FFN hidden gradient: du = drop(dh) ∘ gelu'(u) (gemm.hip's fused epilogue, same helpers) + the column sums of du over the
        // wave's 16 rows as one partial row of cs_part per 16-row block (16 partial rows per 256 rows, as gemm.hip leaves them:
        // launch_reduce_slabs(cs_part, N, ceil(M / 256) * 16, ...) finishes the bias gradient); rows past M contribute zeros
        const uint64_t key = stlt_drop_key(a.epi.dr, a.epi.site);
        const bool row_ok = row < a.M;
        const int srow = row_ok ? row : 0;
        const uint64_t drow = a.epi.drop_rows ? (uint64_t)a.epi.drop_rows[srow] : (uint64_t)srow;
        const float* urow = a.R + (int64_t)srow * a.ldr + col0;
        float* yrow = a.Y + (int64_t)srow * a.ldy + col0;
        const int blk = tm * RB + rb;  // this wave's 16-row block of the matrix
        float* cs_row = a.epi.cs_part + (size_t)blk * (size_t)a.N + col0;
        // the reduction reads ceil(M / 256) * 16 partial rows: the blocks between the matrix's last tile and the end of its 256-row group are zeroed by the last tile row's waves
        const int blk_end = ((a.tiles_m * RB + 15) / 16) * 16;
        const bool last_tm = tm == a.tiles_m - 1;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const bool col_ok = col0 + 16 * t < a.N;
          f32x4 val = {0.f, 0.f, 0.f, 0.f};
          if (row_ok && col_ok) {
            val = gelu_bwd4(acc[t], *reinterpret_cast<const f32x4*>(urow + 16 * t), a.epi, key, drow * (uint64_t)a.N + (uint64_t)(col0 + 16 * t));
            *reinterpret_cast<f32x4*>(yrow + 16 * t) = val;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {  // over the 16 lanes that share lane >> 4 (= the wave's 16 rows)
            float x = val[j];
            x += __shfl_xor(x, 1, 64);
            x += __shfl_xor(x, 2, 64);
            x += __shfl_xor(x, 4, 64);
            x += __shfl_xor(x, 8, 64);
            val[j] = x;
          }
          if (eli == 0 && col_ok) {
            *reinterpret_cast<f32x4*>(cs_row + 16 * t) = val;
            if (last_tm)
              for (int b = a.tiles_m * RB + rb; b < blk_end; b += RB)
                *reinterpret_cast<f32x4*>(a.epi.cs_part + (size_t)b * (size_t)a.N + col0 + 16 * t) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      } else if constexpr (ACT == STLT_ACT_GELU_KEEP) {
        // the training forward's FFN1: u to the add-source pointer, h = drop(gelu(u)) to the output (launch_gelu_fwd's mask indices)
        if (row < a.M) {
          float* urow = const_cast<float*>(a.R) + (int64_t)row * a.ldr + col0;
          float* yrow = a.Y + (int64_t)row * a.ldy + col0;
          const uint64_t key = stlt_drop_key(a.epi.dr, a.epi.site);
          const uint64_t drow = a.epi.drop_rows ? (uint64_t)a.epi.drop_rows[row] : (uint64_t)row;
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            if (col0 + 16 * t < a.N) {
              const f32x4 u = acc[t];
              *reinterpret_cast<f32x4*>(urow + 16 * t) = u;
              f32x4 o = {gelu_epilogue(u[0]), gelu_epilogue(u[1]), gelu_epilogue(u[2]), gelu_epilogue(u[3])};
              if (a.epi.dr.thr) {
                const uint64_t idx0 = drow * (uint64_t)a.N + (uint64_t)(col0 + 16 * t);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = stlt_keep_k(a.epi.dr.thr, key, idx0 + j) ? o[j] * a.epi.dr.scale : 0.f;
              }
              *reinterpret_cast<f32x4*>(yrow + 16 * t) = o;
            }
          }
        }
      } else if (row < a.M) {
        float* yrow = a.Y + (int64_t)row * a.ldy + col0;
        if (ADD) {
          const float* rrow = a.R + (int64_t)row * a.ldr + col0;
          f32x4 rv[NTW];
#pragma unroll
          for (int t = 0; t < NTW; ++t) rv[t] = (col0 + 16 * t < a.N) ? *reinterpret_cast<const f32x4*>(rrow + 16 * t) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[t] += rv[t];
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          f32x4 o = acc[t];
          if (ACT == STLT_ACT_GELU) { o[0] = gelu_epilogue(o[0]); o[1] = gelu_epilogue(o[1]); o[2] = gelu_epilogue(o[2]); o[3] = gelu_epilogue(o[3]); }
          if (ACT == STLT_ACT_RELU) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
          if (STLT_G16_ABLATE & 8) asm volatile("" :: "v"(o));  // timing build: the product stays alive, nothing is stored
          else if (col0 + 16 * t < a.N) *reinterpret_cast<f32x4*>(yrow + 16 * t) = o;  // N % 4 == 0: a 4-column group is inside or outside as a whole
        }
      }
    }
    ++c_it;
    c_kt = 0;
    if (step + 1 < total_steps) init_acc(c_it);  // next tile: accumulators from its bias strip (published by an earlier barrier)
  };

  __builtin_amdgcn_s_barrier();  // the loaders' counted wait + this barrier publish step 0 (DEEP: and step 1) and the first bias strip
  init_acc(0);
  if constexpr (DEEP) {
    struct Set { f32x4 x[2]; f32x4 w[2][NTW]; };  // a whole k-step of fragments: both k-chunks
    // read item q of a k-step's 2 (1 + NTW) fragments: chunk c = q / (1 + NTW), then x (i = 0) or column tile i - 1
    auto read_item = [&](const float* sp, int q, Set& r) {
      const int c = q / (1 + NTW), i = q - c * (1 + NTW);
      if (WKN) {  // k order 16 c + lg + 4 e for MFMA e: element lg of X chunk 4 c + e; row 16 c + lg + 4 e of the [k][n] W image
        if (i == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) r.x[c][e] = sp[x_row + (((4 * c + e) ^ sw) * 4) + lg];
        } else {
          const float* wk = sp + QM * QK + (16 * c + lg) * BN + cg * NTW * 16 + li;
#pragma unroll
          for (int e = 0; e < 4; ++e) r.w[c][i - 1][e] = wk[4 * e * BN + (i - 1) * 16];
        }
      } else {
        const int off = ((4 * c + lg) ^ sw) * 4;
        if (i == 0) r.x[c] = *reinterpret_cast<const f32x4*>(sp + x_row + off);
        else r.w[c][i - 1] = *reinterpret_cast<const f32x4*>(sp + w_row + (i - 1) * 16 * QK + off);
      }
    };
    auto read_set = [&](int st, Set& r) {
#pragma unroll
      for (int q = 0; q < 2 * (1 + NTW); ++q) read_item(smem + st * STAGE, q, r);
    };
    auto mfma_group = [&](const Set& r, int g) {  // group g = (k-chunk g / 4, MFMA g % 4 of the chunk) on every column tile
      const int c = g >> 2, e = g & 3;
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.w[c][t][e], r.x[c][e], acc[t], 0, 0, 0);
    };
    Set A, B;
    read_set(0, A);
    // One k-step: the 8 MFMA groups of `cur` with the read requests of step + 1 spread behind the first four of them (a burst of all
    // 2 (1 + NTW) requests from eight lock-stepped waves fills the LDS queue and holds the waves' MFMAs back behind their own requests),
    // then the wait and the barrier.  The order is pinned (sched_barrier): left alone, hipcc moves the wait and the barrier up in front of the
    // MFMAs (register-only instructions to it) and the requests down to their first use.
    constexpr int NRD = 2 * (1 + NTW), RPG = (NRD + 3) / 4;  // requests per k-step, per MFMA group
    auto kstep = [&](const Set& cur, Set& nxt, int step) {
      const int next_stage = stage + 1 == Q_NSTAGE ? 0 : stage + 1;
      const float* sp = smem + next_stage * STAGE;  // step + 1, published by the previous barrier (after the very last step: a dead read)
      const bool reads = !((STLT_G16_ABLATE & 2) && step > 0);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        mfma_group(cur, g);  // group 0 in front of the first requests: the compiler's wait for `cur` is an lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        if (g < 4 && reads) {
#pragma unroll
          for (int q = g * RPG; q < (g + 1) * RPG && q < NRD; ++q) read_item(sp, q, nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // step + 1's fragments are in registers
      if (!(STLT_G16_ABLATE & 4)) __builtin_amdgcn_s_barrier();  // publishes step + 2
      stage = next_stage;
      if (++c_kt == nk) tile_done(step);
    };
    if (STLT_G16_ABLATE & 2) read_set(1 % Q_NSTAGE, B);
    for (int step = 0; step < total_steps; step += 2) {
      kstep(A, B, step);
      if (step + 1 < total_steps) kstep(B, A, step + 1);
    }
    return;
  }
  read_phase(0, 0, F[0]);
  if (STLT_G16_ABLATE & 2) read_phase(0, 1, F[1]);
  for (int step = 0; step < total_steps; ++step) {
    const int next_stage = stage + 1 == Q_NSTAGE ? 0 : stage + 1;
    const bool ablate_reads = (STLT_G16_ABLATE & 2) && step > 0;
#pragma unroll
    for (int p = 0; p + 1 < NP; ++p) {
      // the phase's first MFMA group goes in front of the next phase's read requests: the compiler's wait for this phase's fragments
      // is an lgkmcnt(0) (the counter is shared with scalar loads), which behind the new requests would wait for those as well
      mfma_phase(p, F[p & 1], 0, 1);
      G16_PIN();
      if (!ablate_reads) read_phase(stage, p + 1, F[(p + 1) & 1]);
      G16_PIN();
      mfma_phase(p, F[p & 1], 1, 4);
      G16_PIN();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own fragment reads of this stage are done
    if (!(STLT_G16_ABLATE & 4)) __builtin_amdgcn_s_barrier();  // retire the stage; step+1 landed
    // the next stage's first fragments (after the last k-step of a tile: the next tile's; after the very last step: a dead read), under the last phase's MFMAs
    if (!ablate_reads) read_phase(next_stage, 0, F[0]);
    G16_PIN();
    mfma_phase(NP - 1, F[1], 0, 4);
    G16_PIN();
    stage = next_stage;
    if (++c_kt == nk) tile_done(step);
  }
#undef G16_PIN
}

template <int RB, int NT, int ACT, bool ADD, bool WKN>
int launch16_as(const Gemm16Args& a, hipStream_t s) {
  static StltPerDeviceOnce attr_done;
  constexpr int SMEM = q_smem_floats<RB, NT>() * (int)sizeof(float);
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  if (!attr_done.flag()) {
    if (hipError_t e = hipFuncSetAttribute((const void*)gemm16_kernel<RB, NT, ACT, ADD, WKN>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); e != hipSuccess)
      return stlt_set_error((int)e, "gemm16: %s", hipGetErrorString(e));
    attr_done.flag() = true;
  }
  const int64_t n_tiles = (int64_t)a.tiles_m * a.tiles_n;
  int64_t G = stlt_device_cus();
  if (G > n_tiles) G = n_tiles;
  hipLaunchKernelGGL((gemm16_kernel<RB, NT, ACT, ADD, WKN>), dim3((unsigned)G), dim3(Q_THREADS), SMEM, s, a);
  return stlt_check_launch("gemm16_kernel");
}

template <int RB, int NT>
int launch16_nt(const Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s) {
  if (wkn && act == STLT_ACT_GELU_BWD) return launch16_as<RB, NT, STLT_ACT_GELU_BWD, false, true>(a, s);
  if (act == STLT_ACT_GELU_BWD) return launch16_as<RB, NT, STLT_ACT_GELU_BWD, false, false>(a, s);  // the same epilogue on the forward build: dX through a transposed weight copy (wt_cache.hip)
  if (wkn) return add ? launch16_as<RB, NT, STLT_ACT_NONE, true, true>(a, s) : launch16_as<RB, NT, STLT_ACT_NONE, false, true>(a, s);
  if (add) return launch16_as<RB, NT, STLT_ACT_NONE, true, false>(a, s);
  if (act == STLT_ACT_GELU_KEEP) return launch16_as<RB, NT, STLT_ACT_GELU_KEEP, false, false>(a, s);
  if (act == STLT_ACT_GELU) return launch16_as<RB, NT, STLT_ACT_GELU, false, false>(a, s);
  if (act == STLT_ACT_RELU) return launch16_as<RB, NT, STLT_ACT_RELU, false, false>(a, s);
  return launch16_as<RB, NT, STLT_ACT_NONE, false, false>(a, s);
}

}  // namespace g16

// per-height dispatchers (one translation unit each); nt must be one of the height's tile widths (stlt_gemm16_tile_ok)
int launch16_rb8(int nt, const g16::Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s);  // gemm16.hip: 128 rows x 16 {3,4,6,8,9,12}
int launch16_rb4(int nt, const g16::Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s);  // gemm16_rb4.hip: 64 rows x 16 {4,6,8,10,12,16}
int launch16_rb2(int nt, const g16::Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s);  // gemm16_rb2.hip: 32 rows x 16 {8,12,16}
