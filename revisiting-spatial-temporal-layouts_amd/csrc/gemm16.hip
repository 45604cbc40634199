// nn.Linear forward  Y = act(X·Wᵀ + b) (+ R)  for UNDER-FILLED launches: few rows (the temporal tower at the reference's default batch
// of 64 clips: M = 2048; the fusion models' 2048 / 2112-row blocks), where gemm.hip's 256 x 128 tiles are fewer than the CUs and the
// launch has to split its contraction over workgroups (stream-K: two 128-KB partial tiles per workgroup written and re-read, a
// fix-up launch, ~25 us of fixed cost on products of 40-100 us: profiles/round3_gemm_train_shapes_b64.txt, 0.32-0.67 of the MFMA peak).
//
// Here the tile is 128 rows x (16 NT) columns, NT in {3, 4, 6, 8, 9, 12}, chosen per launch so that the number of WHOLE tiles is
// close to a multiple of the CU count (M = 2048: N = 768 -> 48-wide tiles = 16 x 16; N = 2304 -> 144 = 16 x 16; N = 3072 -> 192 =
// 16 x 16): every workgroup owns complete outputs, no partial tiles, no second launch.  Narrow tiles fetch more operand bytes per
// FLOP ((128 + 48) x 128 B per k-step against 0.39 MFLOP: ~15 B/clk, above the CU's ~10 B/clk), so they run fetch-bound at ~0.7 of the
// tile's MFMA rate — still well ahead of the split launch they replace.
//
// Structure = mhsa.hip's product phase: v_mfma_f32_16x16x4_f32, 8 MFMA waves each owning one 16-row block and all NT column tiles
// (transposed accumulators: lane = row, registers = 4 consecutive columns -> 16-byte stores), 4 DMA-only loader waves two to five
// k-steps ahead (LDS-DMA with the source-side bank swizzle, three to six stages, counted vmcnt, one barrier per k-step), bias as the accumulators'
// initial value from LDS strips DMA'd in front of each tile's first k-step, persistent workgroups over XCD-contiguous tile ranges (column tile fastest, so
// that the workgroups of an XCD share X row panels).  NT (forward) layout only; K % 32 == 0, N % 4 == 0.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int QM = 128, QK = 32;
constexpr int Q_WAVES = 8, Q_LOADERS = 4;
constexpr int Q_THREADS = 64 * (Q_WAVES + Q_LOADERS);
// operand stages: as many as fit ~150 KB, at most 6.  A narrow tile's k-step is short (48 columns: ~0.7 us), so the loaders must run
// more than two k-steps ahead to cover a miss in the L2 (measured with three stages: ~7 us lost per tile)
template <int NT> constexpr int q_nstage() { return 150 * 1024 / ((128 + 16 * NT) * 32 * 4) > 6 ? 6 : 150 * 1024 / ((128 + 16 * NT) * 32 * 4); }

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

struct Gemm16Args {
  const float* X; const float* W; const float* bias; const float* R;
  float* Y;
  int64_t ldx, ldw, ldr, ldy;
  int M, N, K, tiles_m, tiles_n;
  StltGemmEpi epi;  // ACT == STLT_ACT_GELU_BWD only (R = the pre-activation u, not added)
};

template <int NT> constexpr int q_stage_floats() { return (QM + 16 * NT) * QK; }
constexpr int Q_BIAS_STRIPS = 4;  // see dma_bias
template <int NT> constexpr int q_smem_floats() { return q_nstage<NT>() * q_stage_floats<NT>() + Q_BIAS_STRIPS * 16 * NT; }

// WKN = false: W is (N, K) row-major (nn.Linear's weight; forward products).  WKN = true: W is (K, N) row-major — the input-gradient
// product dX = dY·W of a Linear whose weight (n_out, k_in) is read as it lies, contraction over its rows: the W image in LDS is then
// [32 k][BN n], both operands' fragments are gathered with the k order (16 c + lg + 4 e) so that a lane's four values of a column tile
// are four rows of that image (ds_read_b32; bank-conflict-free for 48 / 144 columns, 2-way otherwise) — no transposed weight copy.
template <int NT, int ACT, bool ADD, bool WKN>
__global__ __launch_bounds__(Q_THREADS, 3) void gemm16_kernel(const Gemm16Args a) {
  constexpr int BN = 16 * NT;
  constexpr int STAGE = q_stage_floats<NT>();
  constexpr int Q_NSTAGE = q_nstage<NT>();
  constexpr int LA = Q_NSTAGE - 1;  // k-steps the loaders run ahead of the MFMA waves
  constexpr int NB_INSTR = 2 * NT;  // 8-row LDS-DMA instructions of the W image per k-step
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
    // ---- loader waves.  X image: loader Ld issues rows [32 Ld, 32 Ld + 32) (4 instructions of 8 rows).  W image (BN rows): the
    // 2 NT instructions are dealt round-robin, loader Ld taking j = Ld, Ld + 4, ...
    const int Ld = wave - Q_WAVES;
    const int drow = lane >> 3, dslot = lane & 7;
    constexpr int NBL_MAX = (NB_INSTR + 3) / 4;
    const int nbl = (NB_INSTR - Ld + 3) / 4;  // this loader's W instructions
    const float* pa[4];
    const float* pb[NBL_MAX];
    auto set_item = [&](int it) {
      int tm, tn;
      item_of(it, tm, tn);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = Ld * 32 + i * 8 + drow;
        int gm = tm * QM + r;
        gm = gm < a.M ? gm : a.M - 1;  // rows past the matrix re-read the last row; their outputs are never stored
        pa[i] = a.X + (int64_t)gm * a.ldx + (dslot ^ ((r >> 1) & 7)) * 4;
      }
#pragma unroll
      for (int i = 0; i < NBL_MAX; ++i) {
        if (WKN) {  // instruction j = Ld + 4 i fills floats [256 j, 256 j + 256) of the [32 k][BN n] image: lane -> (k, n .. n + 3)
          const int f = ((Ld + 4 * i) * 64 + lane) * 4;
          const int k = f / BN, n = f - k * BN;
          int gn = tn * BN + n;
          gn = gn + 4 <= a.N ? gn : a.N - 4;  // columns past the matrix re-read its last four; their outputs are never stored
          pb[i] = a.W + (int64_t)k * a.ldw + gn;
        } else {
          const int r = (Ld + 4 * i) * 8 + drow;
          int gn = tn * BN + r;
          gn = gn < a.N ? gn : a.N - 1;
          pb[i] = a.W + (int64_t)gn * a.ldw + (dslot ^ ((r >> 1) & 7)) * 4;
        }
      }
    };
    // Bias strip of tile `it` (its accumulators' initial value), issued IN FRONT of the tile's first k-step: the counter is in order, so
    // the wait that publishes that k-step publishes the strip as well.  (Until the end of round 4 the strip was issued one k-step before
    // it was read, behind up to LA - 1 newer steps the counted wait lets stay in flight: a race that was almost always won — the 3 KB of
    // bias are L2-resident — and lost once in a 33 000-row launch of a test run.)  Four strips: the loaders are LA <= 5 steps ahead and a
    // tile has nk >= 2 k-steps, so a strip is rewritten at the earliest 4 nk - LA >= 3 barriers after the MFMA waves read it.
    auto dma_bias = [&](int it) {  // loader 0 (and 1, 2 for wide tiles): BN bias values, 64 per instruction
      if (a.bias && Ld * 64 < BN) {
        int tm, tn;
        item_of(it, tm, tn);
        int gn = tn * BN + Ld * 64 + lane;
        gn = gn < a.N ? gn : a.N - 1;
        if (Ld * 64 + lane < BN)
          __builtin_amdgcn_global_load_lds((glb_void_ptr)(a.bias + gn), (lds_void_ptr)(bias_lds + (it & (Q_BIAS_STRIPS - 1)) * BN + Ld * 64), 4, 0, 0);
      }
    };
    int l_it = 0, l_kt = 0, l_stage = 0;
    auto l_step = [&]() {
      if (l_kt == 0) { set_item(l_it); dma_bias(l_it); }
      float* sa = smem + l_stage * STAGE + (Ld * 32) * QK;
      float* sb = smem + l_stage * STAGE + QM * QK;
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_void_ptr)(pa[i] + l_kt * QK), (lds_void_ptr)(sa + i * 8 * QK), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < NBL_MAX; ++i)
        if (i < nbl) __builtin_amdgcn_global_load_lds((glb_void_ptr)(pb[i] + (WKN ? (int64_t)l_kt * QK * a.ldw : (int64_t)l_kt * QK)),
                                                      (lds_void_ptr)(sb + (Ld + 4 * i) * 8 * QK), 16, 0, 0);
      if (++l_kt == nk) { ++l_it; l_kt = 0; }
      if (++l_stage == Q_NSTAGE) l_stage = 0;
    };
    // in-order counter: once at most the instructions of the newest LA - 1 steps are in flight, everything up to the step the MFMA
    // waves read next has landed, and the bias strip in front of it (a strip among the newer instructions only makes the wait stricter).  The count per step is a per-loader
    // constant (4 + nbl), so the wait is one of two immediates.
    constexpr int WAIT_FULL = (LA - 1) * (4 + NBL_MAX), WAIT_LESS = (LA - 1) * (3 + NBL_MAX);
    static_assert(WAIT_FULL < 64, "vmcnt is a 6-bit counter");
    auto wait_ahead = [&]() {
      if (nbl == NBL_MAX) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAIT_FULL) : "memory");
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
        l_step();
        wait_ahead();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ---- MFMA waves: wave w owns rows [16 w, 16 w + 16) of the tile and all NT column tiles
  const int rb = wave;
  const int li = lane & 15, lg = lane >> 4;
  const int sw = (li >> 1) & 7;
  const int x_row = (rb * 16 + li) * QK;
  const int w_row = (QM + li) * QK;
  constexpr int HT = (NT + 1) / 2;  // column tiles per half-chunk (the second half has NT - HT)
  struct Frags { f32x4 x; f32x4 w[HT]; };
  auto read_frags = [&](int stage, int hc) {  // hc 0..3: k-chunk hc >> 1 (16 k each), column tiles HT (hc & 1) ..
    const float* s = smem + stage * STAGE;
    const int off = ((4 * (hc >> 1) + lg) ^ sw) * 4;
    Frags f;
    if (WKN) {  // k order 16 c + lg + 4 e for MFMA e: element lg of X chunk 4 c + e; row 16 c + lg + 4 e of the [k][n] W image
      const int c = hc >> 1;
#pragma unroll
      for (int e = 0; e < 4; ++e) f.x[e] = s[x_row + (((4 * c + e) ^ sw) * 4) + lg];
      const float* wk = s + QM * QK + (16 * c + lg) * BN + li;
#pragma unroll
      for (int t = 0; t < HT; ++t)
        if (HT * (hc & 1) + t < NT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) f.w[t][e] = wk[4 * e * BN + (HT * (hc & 1) + t) * 16];
        }
      return f;
    }
    f.x = *reinterpret_cast<const f32x4*>(s + x_row + off);
#pragma unroll
    for (int t = 0; t < HT; ++t)
      if (HT * (hc & 1) + t < NT) f.w[t] = *reinterpret_cast<const f32x4*>(s + w_row + (HT * (hc & 1) + t) * 16 * QK + off);
    return f;
  };
  f32x4 acc[NT];
  auto init_acc = [&](int it) {
    if (a.bias) {
      const float* src = bias_lds + (it & (Q_BIAS_STRIPS - 1)) * BN + 4 * lg;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(src + 16 * t);
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto mfma_half = [&](const Frags& f, int half) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < HT; ++t)
        if (HT * half + t < NT) acc[HT * half + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[t][e], f.x[e], acc[HT * half + t], 0, 0, 0);
  };

  __builtin_amdgcn_s_barrier();  // the loaders' counted wait + this barrier publish step 0 and the first bias strip
  init_acc(0);
  int c_it = 0, c_kt = 0, stage = 0;
  Frags fa = read_frags(0, 0), fb;
  for (int step = 0; step < total_steps; ++step) {
    const int next_stage = stage + 1 == Q_NSTAGE ? 0 : stage + 1;
    fb = read_frags(stage, 1);
    mfma_half(fa, 0);
    fa = read_frags(stage, 2);
    mfma_half(fb, 1);
    fb = read_frags(stage, 3);
    mfma_half(fa, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own fragment reads of this stage are done
    __builtin_amdgcn_s_barrier();                        // retire the stage; step+1 landed
    const bool item_done = c_kt + 1 == nk;
    if (!item_done) fa = read_frags(next_stage, 0);
    mfma_half(fb, 1);
    stage = next_stage;
    ++c_kt;
    if (!item_done) continue;

    // ---- epilogue of tile c_it: lane (li, lg) holds columns 16 t + 4 lg .. + 3 of row 16 rb + li
    int tm, tn;
    item_of(c_it, tm, tn);
    {
      int eli = li, elg = lg;  // opaque copies: the address arithmetic is recomputed per tile instead of living in registers across the k-loop
      asm volatile("" : "+v"(eli), "+v"(elg));
      const int row = tm * QM + rb * 16 + eli;
      const int col0 = tn * BN + 4 * elg;
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
        float* cs_row = a.epi.cs_part + (size_t)(tm * 8 + rb) * (size_t)a.N + col0;
        const bool pad_blocks = tm == a.tiles_m - 1 && (tm & 1) == 0;  // a last tile that is the first half of a 256-row group: zero the other half's partial rows
#pragma unroll
        for (int t = 0; t < NT; ++t) {
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
            if (pad_blocks) *reinterpret_cast<f32x4*>(cs_row + (size_t)8 * a.N + 16 * t) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      } else if (row < a.M) {
        float* yrow = a.Y + (int64_t)row * a.ldy + col0;
        if (ADD) {
          const float* rrow = a.R + (int64_t)row * a.ldr + col0;
          f32x4 rv[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) rv[t] = (col0 + 16 * t < a.N) ? *reinterpret_cast<const f32x4*>(rrow + 16 * t) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] += rv[t];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x4 o = acc[t];
          if (ACT == STLT_ACT_GELU) { o[0] = gelu_epilogue(o[0]); o[1] = gelu_epilogue(o[1]); o[2] = gelu_epilogue(o[2]); o[3] = gelu_epilogue(o[3]); }
          if (ACT == STLT_ACT_RELU) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
          if (col0 + 16 * t < a.N) *reinterpret_cast<f32x4*>(yrow + 16 * t) = o;  // N % 4 == 0: a 4-column group is inside or outside as a whole
        }
      }
    }
    ++c_it;
    c_kt = 0;
    if (step + 1 < total_steps) {  // next tile: accumulators from its bias strip (published by the last k-step's barrier), first fragments
      init_acc(c_it);
      fa = read_frags(stage, 0);
    }
  }
}

template <int NT, int ACT, bool ADD, bool WKN>
int launch16_as(const Gemm16Args& a, hipStream_t s) {
  static StltPerDeviceOnce attr_done;
  constexpr int SMEM = q_smem_floats<NT>() * (int)sizeof(float);
  if (!attr_done.flag()) {
    if (hipError_t e = hipFuncSetAttribute((const void*)gemm16_kernel<NT, ACT, ADD, WKN>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); e != hipSuccess)
      return stlt_set_error((int)e, "gemm16: %s", hipGetErrorString(e));
    attr_done.flag() = true;
  }
  const int64_t n_tiles = (int64_t)a.tiles_m * a.tiles_n;
  int64_t G = stlt_device_cus();
  if (G > n_tiles) G = n_tiles;
  hipLaunchKernelGGL((gemm16_kernel<NT, ACT, ADD, WKN>), dim3((unsigned)G), dim3(Q_THREADS), SMEM, s, a);
  return stlt_check_launch("gemm16_kernel");
}

template <int NT>
int launch16_nt(const Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s) {
  if (wkn && act == STLT_ACT_GELU_BWD) return launch16_as<NT, STLT_ACT_GELU_BWD, false, true>(a, s);
  if (wkn) return add ? launch16_as<NT, STLT_ACT_NONE, true, true>(a, s) : launch16_as<NT, STLT_ACT_NONE, false, true>(a, s);
  if (add) return launch16_as<NT, STLT_ACT_NONE, true, false>(a, s);
  if (act == STLT_ACT_GELU) return launch16_as<NT, STLT_ACT_GELU, false, false>(a, s);
  if (act == STLT_ACT_RELU) return launch16_as<NT, STLT_ACT_RELU, false, false>(a, s);
  return launch16_as<NT, STLT_ACT_NONE, false, false>(a, s);
}

int launch16_any(int nt, const Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s) {
  switch (nt) {
    case 3: return launch16_nt<3>(a, act, add, wkn, s);
    case 4: return launch16_nt<4>(a, act, add, wkn, s);
    case 6: return launch16_nt<6>(a, act, add, wkn, s);
    case 8: return launch16_nt<8>(a, act, add, wkn, s);
    case 9: return launch16_nt<9>(a, act, add, wkn, s);
    default: return launch16_nt<12>(a, act, add, wkn, s);
  }
}

constexpr int NT_CHOICES[] = {3, 4, 6, 8, 9, 12};

// Launch-time estimates (us), fitted to stand-alone measurements on MI355X (profiles/round4_gemm16_shapes.txt).
// Small tiles: rounds x (k-steps x the tile's MFMA time at 0.54 TFLOP/s per CU + ~7 us per tile for prologue, epilogue and the
// launch boundary) — the operand fetch is not the limit even at 48 columns.
// The WKN build (input gradients: W read as [k][n]) gathers its fragments with ds_read_b32 and measures 1.25 - 1.4x the k-step time.
double est16_us(int64_t M, int64_t N, int64_t K, int nt, int64_t cus, bool wkn = false) {
  const int64_t tiles = ((M + QM - 1) / QM) * ((N + 16 * nt - 1) / (16 * nt));
  const int64_t rounds = (tiles + cus - 1) / cus;
  // the input-gradient build gathers its [k][n] fragments with ds_read_b32: x 1.38 per k-step on narrow tiles, less on wide ones (fitted to
  // the *_dx rows of profiles/round4_gemm16_shapes.txt)
  const double wkn_cost = nt <= 4 ? 1.38 : (nt == 6 ? 1.3 : (nt == 8 ? 1.2 : (nt == 9 ? 1.15 : 1.08)));
  const double step = 2.0 * QM * 16.0 * nt * QK / 0.54e6 * (wkn ? wkn_cost : 1.0);
  return (double)rounds * ((double)(K / QK) * step + 7.0) + 1.0;
}
// gemm.hip's launch: 256 x 128 tiles at 3.62 us per k-step; whole-tile rounds when they fill >= 0.9 of the last round, else equal
// k-step shares (stream-K) + the fixed cost of the partial tiles and the fix-up launch (24 us + 0.1 us per tile below one round,
// ~30 us for the tail of a longer launch)
double est_big_us(int64_t M, int64_t N, int64_t K, int64_t cus, bool nn = false) {  // nn: the input-gradient (NN) build, ~6 % slower per k-step
  const int64_t tiles = ((M + 255) / 256) * ((N + 127) / 128);
  const int64_t rounds = (tiles + cus - 1) / cus;
  const double step = nn ? 3.85 : 3.62, nk = (double)(K / QK);
  const double fill = (double)tiles / (double)(rounds * cus);
  if (fill >= 0.9) return (double)rounds * (nk * step + 10.0);
  return (double)tiles * nk / (double)cus * step + (tiles < cus ? 24.0 + 0.1 * (double)tiles : 30.0);
}

}  // namespace

// Would launch_linear hand this product to the small-tile kernel, and with which tile width?  0: no (gemm.hip keeps it).
// STLT_GEMM16=0 switches the kernel off, STLT_GEMM16=1 forces it onto every product it can take (A/B runs); otherwise the two
// launch-time estimates above decide.
static int g_gemm16_mode = -2;  // -2: not read yet; -1: by estimate; 0: off; 1: every product the kernel can take
int stlt_gemm16_set_mode(int mode) {
  if (mode < -2 || mode > 1) return stlt_set_error(STLT_EINVAL, "small-tile products: mode -1 (by estimate), 0 (off), 1 (always) or -2 (back to STLT_GEMM16 / the default)");
  g_gemm16_mode = mode;  // -2: the next routing decision re-reads the environment
  return 0;
}
int stlt_gemm16_choice(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, bool wkn) {
  if (g_gemm16_mode == -2) { const char* e = getenv("STLT_GEMM16"); g_gemm16_mode = e ? (atoi(e) == 0 ? 0 : (atoi(e) == 1 ? 1 : -1)) : -1; }
  const int mode = g_gemm16_mode;
  static const int force_nt = [] { const char* e = getenv("STLT_GEMM16_NT"); return e ? atoi(e) : 0; }();
  if (mode == 0) return 0;
  if (M <= 0 || N <= 0 || K < 2 * QK || K % QK != 0 || N % 4 != 0 || ldx % 4 != 0 || ldw % 4 != 0 || M > 0x3fffff00LL || N > 0x3fffff00LL) return 0;
  const int64_t cus = stlt_device_cus();
  int best = 0;
  double best_us = 1e30;
  for (int nt : NT_CHOICES) {
    if (force_nt && nt != force_nt) continue;
    const int64_t tiles = ((M + QM - 1) / QM) * ((N + 16 * nt - 1) / (16 * nt));
    if (tiles > 0x3fffffffLL) continue;
    const double us = est16_us(M, N, K, nt, cus, wkn);
    if (us < best_us) { best_us = us; best = nt; }
  }
  if (best == 0) return 0;
  if (mode == 1) return best;
  // forward: the small tiles must win by 3 %; input gradient: a tie goes to the small tiles (no fix-up launch beside the side stream's products)
  return best_us < (wkn ? 1.03 : 0.97) * est_big_us(M, N, K, cus, wkn) ? best : 0;
}

// Estimated duration (us) of the nn.Linear forward launch_linear would make for this shape: the faster of the two kernels' estimates
// (what the routing picks).  Used by the fused MHSA dispatch to price the product + attention-core pair it competes with.
double stlt_linear_est_us(int64_t M, int64_t N, int64_t K) {
  const int64_t cus = stlt_device_cus();
  double us = est_big_us(M, N, K, cus);
  const int nt = stlt_gemm16_choice(M, N, K, K, K, false);
  if (nt > 0) { const double small = est16_us(M, N, K, nt, cus); if (small < us) us = small; }
  return us;
}

// Y (M, N; ldy) = act(X (M, K; ldx) · W (N, K; ldw)ᵀ + bias) (+ R (ldr)) on the small-tile kernel; *taken = false when the shape is not its.
int launch_linear_gemm16(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* r, int64_t ldr, float* y, int64_t ldy,
                         int64_t M, int64_t N, int64_t K, int act, hipStream_t s, bool* taken, int force_nt) {
  *taken = false;
  if (act != STLT_ACT_NONE && act != STLT_ACT_GELU && act != STLT_ACT_RELU) return 0;
  if (r && act != STLT_ACT_NONE) return 0;
  int nt = force_nt;
  if (nt == 0) nt = stlt_gemm16_choice(M, N, K, ldx, ldw, false);
  else if (M <= 0 || N <= 0 || K < 2 * QK || K % QK != 0 || N % 4 != 0 || ldx % 4 != 0 || ldw % 4 != 0 || M > 0x3fffff00LL || N > 0x3fffff00LL ||
           (nt != 3 && nt != 4 && nt != 6 && nt != 8 && nt != 9 && nt != 12))
    return stlt_set_error(STLT_EINVAL, "gemm16: K must be a multiple of 32 (>= 64), N and the row pitches multiples of 4, tile width 16 x {3,4,6,8,9,12}");
  if (nt == 0) return 0;
  if (!x || !w || !y) return stlt_set_error(STLT_EINVAL, "gemm16: null pointer");
  if (ldx < K || ldw < K || ldy < N || (r && ldr < N) || ldy % 4 != 0 || (r && ldr % 4 != 0))
    return stlt_set_error(STLT_EINVAL, "gemm16: bad leading dimension (ldx=%lld ldw=%lld ldy=%lld ldr=%lld)", (long long)ldx, (long long)ldw, (long long)ldy, (long long)ldr);
  Gemm16Args a{};
  a.X = x; a.W = w; a.bias = bias; a.R = r; a.Y = y;
  a.ldx = ldx; a.ldw = ldw; a.ldr = ldr; a.ldy = ldy;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)((M + QM - 1) / QM);
  a.tiles_n = (int)((N + 16 * nt - 1) / (16 * nt));
  if ((int64_t)a.tiles_m * a.tiles_n > 0x3fffffffLL) return stlt_set_error(STLT_EINVAL, "gemm16: too many tiles");
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)M * (double)N * (double)K);
  *taken = true;
  return launch16_any(nt, a, act, r != nullptr, false, s);
}

// C (rows, k_in; ldc) = dY (rows, n_out; ld_dy) · W (n_out, k_in) (+ R): the input gradient of a Linear on the small-tile kernel, W read
// as it lies (WKN build).  *taken = false: the product stays on gemm.hip's NN kernel (shape not taken, or estimated slower).
int launch_input_grad_gemm16(const float* dy, int64_t ld_dy, const float* w, int64_t n_out, int64_t k_in, const float* r, int64_t ldr, float* c,
                             int64_t ldc, int64_t rows, hipStream_t s, bool* taken, int force_nt, const StltGemmEpi* gelu_bwd) {
  // gelu_bwd != null: the FFN hidden gradient in the epilogue — c = drop(dy·w) ∘ gelu'(r) with r the pre-activation (not added), and the
  // column sums of c left in gelu_bwd->cs_part (16 partial rows per 256 rows; train.hip: stlt_ffn_hidden_backward_fused)
  *taken = false;
  if (gelu_bwd && (!r || !gelu_bwd->cs_part)) return stlt_set_error(STLT_EINVAL, "gemm16 (GELU backward): the pre-activation and a column-sum buffer are required");
  int nt = force_nt;
  if (nt == 0) nt = stlt_gemm16_choice(rows, k_in, n_out, ld_dy, k_in, true);
  else if (rows <= 0 || k_in <= 0 || n_out < 2 * QK || n_out % QK != 0 || k_in % 4 != 0 || ld_dy % 4 != 0 || rows > 0x3fffff00LL || k_in > 0x3fffff00LL ||
           (nt != 3 && nt != 4 && nt != 6 && nt != 8 && nt != 9 && nt != 12))
    return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): n_out must be a multiple of 32 (>= 64), k_in and the row pitches multiples of 4");
  if (nt == 0) return 0;
  if (!dy || !w || !c) return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): null pointer");
  if (ld_dy < n_out || ldc < k_in || (r && ldr < k_in) || ldc % 4 != 0 || (r && ldr % 4 != 0)) return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): bad leading dimension");
  Gemm16Args a{};
  if (gelu_bwd) a.epi = *gelu_bwd;
  a.X = dy; a.W = w; a.bias = nullptr; a.R = r; a.Y = c;
  a.ldx = ld_dy; a.ldw = k_in; a.ldr = ldr; a.ldy = ldc;
  a.M = (int)rows; a.N = (int)k_in; a.K = (int)n_out;
  a.tiles_m = (int)((rows + QM - 1) / QM);
  a.tiles_n = (int)((k_in + 16 * nt - 1) / (16 * nt));
  if ((int64_t)a.tiles_m * a.tiles_n > 0x3fffffffLL) return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): too many tiles");
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)rows * (double)k_in * (double)n_out);
  *taken = true;
  return launch16_any(nt, a, gelu_bwd ? STLT_ACT_GELU_BWD : STLT_ACT_NONE, r != nullptr && !gelu_bwd, true, s);
}
