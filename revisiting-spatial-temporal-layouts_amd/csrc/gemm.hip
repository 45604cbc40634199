// nn.Linear forward  Y = act(X · Wᵀ + b)  on the f32-input matrix cores of gfx950.
//
//   X (M,K) row-major with leading dim ldx, W (N,K) row-major (torch (out,in)), Y (M,N) with ldy.
//   Both operands are K-contiguous, so A- and B-fragments have the same shape: lane (r = lane&31,
//   h = lane>>5) of a wave reads 16 B = 4 consecutive k of row r (ds_read_b128) and feeds them to four
//   v_mfma_f32_32x32x2_f32.  The k index inside an 8-wide chunk is permuted (MFMA e pairs k = 8c+e with
//   k = 8c+4+e); a sum over k does not care, and A and B use the same permutation.
//
// Structure (one persistent 512-thread workgroup per CU = 8 waves as 4x2):
//   * block tile 256x128x32, each wave 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  The per-CU
//     fetch path (~10 B/clk measured) is co-critical with the f32 MFMA pipe: two independent 128x128 tiles per
//     CU need 8 B/clk at full MFMA rate and queue up at VMEM issue; one 256x128 tile needs 6 B/clk;
//   * a workgroup walks its tiles (XCD-contiguous tile order) as ONE flattened stream of k-steps, so the next
//     tile's operands are already in flight while a tile's epilogue runs;
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write.  The LDS image is
//     lane-linear per wave instruction (8 rows x 128 B), so the bank swizzle chunk ^= (row>>1)&7 is applied
//     to the per-lane SOURCE address and again on the fragment reads (conflict-free ds_read_b128);
//   * three LDS stages (144 KB): step s+2's DMA is issued in thirds between the MFMA chunks of step s (into the
//     stage retired by the previous barrier); the barrier that retires step s is preceded by a COUNTED
//     vmcnt(6) that only waits for step s+1's data;
//   * fragments are read one 8-wide k-chunk ahead (ping-pong registers), and the read of the next step's first
//     chunk sits between the barrier and the last 16 MFMAs of the current step;
//   * the MFMA computes the TRANSPOSED tile (W fragment as the A operand, X fragment as B): a lane then owns one
//     output row and its 16 accumulator registers are 4 groups of 4 CONSECUTIVE output columns, so the epilogue is
//     16-byte stores (16 per wave and tile instead of 64 scalar ones: the store tail is issue-bound, it cost 4 % at
//     K = 768); products and summation order are unchanged, results are bit-identical to the untransposed form;
//   * the bias is the accumulators' initial value, fetched one tile ahead (read from its LDS strip straight into
//     the accumulator registers when a tile's epilogue re-initialises them).
//
// Wave specialisation (template flag WS, the default for launches that are not stream-K): the workgroup has four more
// waves (8..11, one per SIMD) that do nothing but issue the LDS-DMA, and the eight MFMA waves issue none.  The CU's
// load path delivers ~10 B/clk and a k-step needs 6: a wave that issues a DMA queues behind the other waves' pieces
// (~130 cycles per instruction, in order, so its MFMAs wait too), and because SIMD partners alternate whole 16-MFMA
// runs both partners reach their DMA issue together and the matrix pipe idles meanwhile.  Ablation (timing-only
// builds, 229376x2304x768 / 229376x768x3072): no DMA issue 131 -> 144 / 136 -> 146 TFLOP/s, no fragment reads +-0, no
// barrier -1 %, no counted wait +-1 %, no epilogue stores +4 % / +1 %.  A loader wave parks at the barrier between
// its bursts and costs the matrix pipe nothing.
//
// Operand layouts (template flags) for the backward pass of nn.Linear:
//   TA=0: A stored (M, Kc) k-contiguous      TA=1: A stored (Kc, M) m-contiguous  (dW = dYᵀ·X reads dY this way)
//   TB=0: B stored (N, Kc) k-contiguous      TB=1: B stored (Kc, N) n-contiguous  (dX = dY·W reads W this way)
// Contraction-major operands are staged as [32 k][256 m] / [32 k][128 n] images (whole 1-KB / 512-B rows per DMA
// instruction) and their fragments are four conflict-free ds_read_b32.  n_split > 1 splits the contraction range
// over work items; split s writes its partial product to C + s*slab_stride (summed by reduce_slabs_kernel:
// deterministic, no atomics).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

#ifndef STLT_GEMM_PRIO_MODE
#define STLT_GEMM_PRIO_MODE 0  // 0 off (measured best: 134.4 TF); 1 static priority for waves 4-7 (134); 2 SIMD partners alternate priority per half k-step (133.7)
#endif

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int GEMM_WAVES = 8;                           // MFMA waves
constexpr int GEMM_THREADS = 64 * GEMM_WAVES;
constexpr int GEMM_LOADERS = 4;                         // WS: DMA-only waves 8..11
constexpr int GEMM_THREADS_WS = 64 * (GEMM_WAVES + GEMM_LOADERS);
constexpr int NSTAGE = 3;
constexpr int STAGE_FLOATS = (BM + BN) * BK;  // 12288 floats = 48 KB per stage

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

#ifndef STLT_GEMM_ABLATE
#define STLT_GEMM_ABLATE 0  // timing-only builds (wrong results): bit 0 no steady-state DMA, bit 3 loaders do not wait for their DMA, bit 4 no epilogue stores, bit 5 loaders re-read k-step 0 of their first tile (cache-hot source)
#endif
#ifndef STLT_GEMM_RES_PREFETCH
// (requesting the second row half's pieces ahead of the first half's stores as well was tried: 8 pieces spill 72 registers at the 168 budget, 4 spill 32)
#define STLT_GEMM_RES_PREFETCH 1  // the add-source pieces of a tile's first row half are requested at the start of the tile's last k-step (forward layout, loader-wave build, whole tiles): cfg2 / 1024 clips with the residual adds in the epilogues 107.2 -> 106.7 ms (profiles/round3_fwd_residual_ab.txt)
#endif
#ifndef STLT_GEMM_STORE_NT
#define STLT_GEMM_STORE_NT 0  // 1: non-temporal epilogue stores — measured slower (cfg2 / 1024 clips: 107.3 -> 108.1 ms per forward, round 3)
#endif
#ifndef STLT_GEMM_GROUPED
#define STLT_GEMM_GROUPED 1  // 1: every XCD walks a contiguous stretch of a band-major tile order (bands of 4 M-panels, N outer inside a band); 0: round-1 order
#endif
#ifndef STLT_GEMM_WS_DEFAULT
#define STLT_GEMM_WS_DEFAULT 1
#endif
__device__ __forceinline__ float half_wave_sum(float x) {  // over the 32 lanes that share lane >> 5
  x += __shfl_xor(x, 1, 64);
  x += __shfl_xor(x, 2, 64);
  x += __shfl_xor(x, 4, 64);
  x += __shfl_xor(x, 8, 64);
  x += __shfl_xor(x, 16, 64);
  return x;
}

// Stream-K launches on a full grid of 8 x Gx workgroups with at least one round of whole tiles to spare run as a hybrid:
// `dp_rounds` rounds of whole tiles in the XCD-grouped order (no partial tiles, the L2-friendly order), then only the
// remaining tiles — positions [dp_rounds*G, n_tiles) of the same order — as k-step ranges.  XCD x owns the tail steps
// [P[x], P[x+1]), cut into ranges of S[x] steps for its workgroups in turn (per-XCD lengths: room for clock-weighted
// ranges; equal today).  dp_rounds = 0: the plain stream-K assignment (G equal ranges over every tile).
struct SkPlan { int dp_rounds; int P[9]; int S[8]; };
struct SkNone {};
template <bool B, class T> auto karg(const T& t) { if constexpr (B) return t; else return SkNone{}; }

// GROUP (with SK and WS): the launch walks the tiles of several products (StltGemmGroup, by value in the kernel arguments:
// a wave reads the fields of the product its current tile belongs to with scalar loads); X / W / R / Y / M / N / K of the
// single-product form are unused.  Plain stream-K assignment over the concatenated k-step space.
template <int ACT, bool STAMP, bool TA, bool TB, bool ADD, bool SK, bool WS, bool GROUP = false>
__global__ __launch_bounds__(WS ? GEMM_THREADS_WS : GEMM_THREADS, WS ? 3 : 2) void gemm_nt_kernel(const float* __restrict__ X, int64_t ldx,
                                                                  const float* __restrict__ W, int64_t ldw,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ R, int64_t ldr,
                                                                  float* __restrict__ Y, int64_t ldy,
                                                                  int64_t slab_stride, int M, int N, int K,
                                                                  int tiles_m, int tiles_n, int n_split,
                                                                  float* __restrict__ partials,
                                                                  unsigned long long* __restrict__ dbg,
                                                                  const std::conditional_t<SK, SkPlan, SkNone> plan_in,
                                                                  const std::conditional_t<GROUP, StltGemmGroup, SkNone> grp_in,
                                                                  const std::conditional_t<ACT == STLT_ACT_GELU_BWD, StltGemmEpi, SkNone> epi_in) {
  SkPlan plan{}; StltGemmGroup grp{}; StltGemmEpi epi{};
  if constexpr (SK) plan = plan_in;
  if constexpr (GROUP) grp = grp_in;
  if constexpr (ACT == STLT_ACT_GELU_BWD) epi = epi_in;
  static_assert(!GROUP || (SK && WS), "grouped launches are stream-K launches of the loader-wave build");
  static_assert(ACT != STLT_ACT_GELU_BWD || (ADD && !GROUP), "the fused GELU backward reads u through the add-source");
  constexpr int prio = STLT_GEMM_PRIO_MODE;
  constexpr int NBIAS = 2;  // bias strips, by tile parity
  __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE_FLOATS + NBIAS * BN];  // operand stages + bias strips

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64x64
  const int lr = lane & 31, lh = lane >> 5;
  const int nk_total = K / BK;
  int nk = GROUP ? 1 : nk_total / n_split;    // k-steps per work item (launcher guarantees divisibility); GROUP: of the wave's current tile
  const int n_tiles = GROUP ? grp.tile_base[grp.n] : tiles_m * tiles_n * n_split;  // work items: (output tile, contraction split), split fastest

  // XCD-contiguous tile order: workgroups b and b+8 share an XCD (round-robin dispatch), so virtual id
  // v = (b%8)*(G/8) + b/8 gives each XCD a contiguous run of tiles (N fastest: they share the X panel).
  const int G = gridDim.x;
  int v = blockIdx.x;
  if ((G & 7) == 0) v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  // Work assignment.  Default: whole work items v, v+G, v+2G, ...  Stream-K (SK): the launch's k-steps
  // (tile-major) are cut into G equal contiguous ranges, so a launch with fewer tiles than CUs, or a ragged last
  // round, still keeps every CU busy; a range may begin and/or end inside a tile, and those segments go to the
  // partial-tile slots 2v (range begins inside / at this tile) and 2v+1 (range ends inside it), which
  // gemm_fixup_kernel sums in workgroup order (deterministic).
  // Grouped order (whole-tile launches on a grid that is a multiple of 8): the 8 XCDs have private L2s, so each XCD walks
  // its own contiguous stretch of a band-major order — bands of GROUP_M = 4 M-panels, inside a band the N-panel index
  // outermost.  The Gx = G/8 workgroups of an XCD hold Gx consecutive positions per round = 4 M-panels x Gx/4 N-panels
  // (4 X panels + 8 W panels instead of ~2 + all of them), and the next round moves on along N inside the same band, so the
  // band's X panels stay in the XCD's L2 while its W panels stream.
  constexpr int GROUP_M = 4;
  const bool grouped = STLT_GEMM_GROUPED && !SK && n_split == 1 && (G & 7) == 0;
  const int g_xcd = blockIdx.x & 7, g_local = blockIdx.x >> 3, g_gx = G >> 3;
  const int g_rounds = (n_tiles + G - 1) / G;
  int my_tiles, sk_first = 0, sk_kt0 = 0, sk_tail = 0;
  const bool hy = SK && plan.dp_rounds > 0;  // hybrid: whole-tile rounds, then a stream-K tail
  const int hy_R = hy ? plan.dp_rounds : 0;
  int hy_steps = 0;
  int g_p0 = 0, g_first = 0, g_nk_last = 1, g_total = 0;  // GROUP: product of the first tile, first global tile, nk of the last tile, steps
  if constexpr (GROUP) {
    const int total = grp.step_base[grp.n];
    const int S = (total + G - 1) / G;
    const int s0 = v * S;
    if (s0 >= total) return;
    const int s1 = s0 + S < total ? s0 + S : total;
    while (grp.step_base[g_p0 + 1] <= s0) ++g_p0;
    const int nk0 = grp.p[g_p0].nk;
    const int lt0 = (s0 - grp.step_base[g_p0]) / nk0;
    sk_kt0 = s0 - grp.step_base[g_p0] - lt0 * nk0;
    g_first = grp.tile_base[g_p0] + lt0;
    int p1 = g_p0;
    while (grp.step_base[p1 + 1] <= s1 - 1) ++p1;
    g_nk_last = grp.p[p1].nk;
    const int lt1 = (s1 - 1 - grp.step_base[p1]) / g_nk_last;
    my_tiles = grp.tile_base[p1] + lt1 - g_first + 1;
    sk_tail = s1 - (grp.step_base[p1] + lt1 * g_nk_last);  // end offset inside the last tile (== its nk when the range ends on a tile boundary)
    g_total = s1 - s0;
  } else
  if (hy) {
    const int e = plan.P[g_xcd + 1];
    const int s0 = plan.P[g_xcd] + g_local * plan.S[g_xcd];
    const int s1 = s0 + plan.S[g_xcd] < e ? s0 + plan.S[g_xcd] : e;
    int sk_tiles = 0;
    if (s0 < s1) {
      sk_first = s0 / nk;
      sk_kt0 = s0 - sk_first * nk;
      const int last = (s1 - 1) / nk;
      sk_tiles = last - sk_first + 1;
      sk_tail = s1 - last * nk;
      hy_steps = s1 - s0;
    }
    my_tiles = hy_R + sk_tiles;
  } else if (SK) {
    const int total = n_tiles * nk;
    const int S = (total + G - 1) / G;
    const int s0 = v * S;
    if (s0 >= total) return;
    const int s1 = s0 + S < total ? s0 + S : total;
    sk_first = s0 / nk;
    sk_kt0 = s0 - sk_first * nk;
    const int last = (s1 - 1) / nk;
    my_tiles = last - sk_first + 1;
    sk_tail = s1 - last * nk;  // == nk when the range ends on a tile boundary
  } else if (grouped) {
    // position of this workgroup's it-th tile in the band-major order: XCD x owns positions [x*R*Gx, (x+1)*R*Gx)
    my_tiles = 0;
    const int p0 = g_xcd * g_rounds * g_gx + g_local;
    if (p0 < n_tiles) my_tiles = (n_tiles - p0 + g_gx - 1) / g_gx;
    if (my_tiles > g_rounds) my_tiles = g_rounds;
    if (my_tiles <= 0) return;
  } else {
    my_tiles = (n_tiles - v + G - 1) / G;  // tiles v, v+G, v+2G, ...
    if (my_tiles <= 0) return;
  }
  if (dbg && tid == 0) {  // diagnostics only (tools/gemm_block_times.py): per-workgroup start/end on the 100 MHz clock
    dbg[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
    dbg[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID[3:0]
    dbg[4 * blockIdx.x + 3] = my_tiles;
    dbg[4 * (size_t)gridDim.x + (size_t)gridDim.x * GEMM_WAVES * 6 + 1024 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime();  // shader clock, behind the stamp / trace regions
  }
  const int total_steps = GROUP ? g_total : hy ? hy_R * nk + hy_steps : SK ? (my_tiles - 1) * nk + sk_tail - sk_kt0 : my_tiles * nk;
  // does this workgroup compute every k-step of its it-th tile?  (otherwise the segment is a partial: no bias, raw store)
  auto seg_complete = [&](int it) {
    if (!SK) return true;
    if (it < hy_R) return true;
    return (it > hy_R || sk_kt0 == 0) && (it < my_tiles - 1 || sk_tail == (GROUP ? g_nk_last : nk));
  };
  // GROUP: product and local tile of the workgroup's it-th tile (a range covers a few tiles: the scan is short)
  auto group_tile = [&](int it, int& pi, int& lt) {
    const int gt = g_first + it;
    pi = g_p0;
    while (grp.tile_base[pi + 1] <= gt) ++pi;
    lt = gt - grp.tile_base[pi];
  };
  // first k-step of the it-th tile: only the first stream-K segment may begin inside a tile
  auto kt_begin = [&](int it) { return (SK && it == hy_R) ? sk_kt0 : 0; };

  auto tile_origin = [&](int it, int& m0, int& n0, int& split) {
    if constexpr (GROUP) {
      int pi, lt;
      group_tile(it, pi, lt);
      const int tn_all = grp.p[pi].tiles_n;
      const int tm = lt / tn_all;
      split = 0;
      m0 = tm * BM;
      n0 = (lt - tm * tn_all) * BN;
      return;
    }
    if (grouped || hy) {
      // hybrid: the whole-tile rounds take positions [0, R*G) XCD by XCD, the tail tiles follow in the same order
      const int p = hy ? (it < hy_R ? (g_xcd * hy_R + it) * g_gx + g_local : hy_R * G + sk_first + (it - hy_R))
                       : (g_xcd * g_rounds + it) * g_gx + g_local;
      const int band = p / (GROUP_M * tiles_n), w = p - band * (GROUP_M * tiles_n);
      const int rows = tiles_m - band * GROUP_M < GROUP_M ? tiles_m - band * GROUP_M : GROUP_M;
      const int tn_g = w / rows;
      split = 0;
      m0 = (band * GROUP_M + (w - tn_g * rows)) * BM;
      n0 = tn_g * BN;
      return;
    }
    const int item = SK ? sk_first + it : v + it * G;
    const int tile = item / n_split;
    split = item - tile * n_split;
    const int tm = tile / tiles_n;
    m0 = tm * BM;
    n0 = (tile - tm * tiles_n) * BN;
  };

  // ---- DMA side -------------------------------------------------------------------------------------
  // Per wave and k-step: 4 instructions for the A stage image, 2 for the B image.  Pointers are computed once
  // per work item; a k-step only adds the operand's k-stride.
  //   k-contiguous operand : image [rows][32 k] (swizzled); an instruction covers 8 rows x 128 B
  //   contraction-major    : image [32 k][256 m | 128 n]; an instruction covers one 1-KB k-row (A) / two 512-B k-rows (B)
  const int drow = lane >> 3, dslot = lane & 7;
  constexpr int NV = WS ? 2 : 1;  // a loader wave does the DMA share of MFMA waves 2j and 2j+1
  // Addresses = a wave-uniform base per operand (the tile's origin: scalar registers; the k offset is added with scalar arithmetic) + a
  // per-lane 32-bit byte offset fixed for the tile, so that a k-step's DMA issue costs no vector-ALU instruction (round 5: with 64-bit
  // per-lane pointers every instruction paid a 64-bit vector add, issued beside the MFMA waves of the loader's SIMD; the small-tile
  // kernel's loaders cost its MFMA waves 4 - 6 % that way, profiles/round5_gemm16_ablation.txt)
  uint32_t voa[NV][4];
  uint32_t vob[NV][2];
  const char* base_a = nullptr;
  const char* base_b = nullptr;
  const int vw0 = WS ? 2 * (wave - GEMM_WAVES) : wave;  // first "virtual wave" whose DMA share this wave issues (WS: loaders only)
  int64_t a_kstep = (TA ? (int64_t)BK * ldx : BK) * (int64_t)sizeof(float);  // bytes per k-step
  int64_t b_kstep = (TB ? (int64_t)BK * ldw : BK) * (int64_t)sizeof(float);
  auto dma_set_tile = [&](int it) {
    int m0, n0, split;
    tile_origin(it, m0, n0, split);
    if constexpr (GROUP) {  // this tile's product (loader waves only: they own nk / the k strides of the DMA stream)
      int pi, lt;
      group_tile(it, pi, lt);
      X = grp.p[pi].a; ldx = grp.p[pi].lda; W = grp.p[pi].b; ldw = grp.p[pi].ldb; M = grp.p[pi].M; N = grp.p[pi].N; nk = grp.p[pi].nk;
      a_kstep = (TA ? (int64_t)BK * ldx : BK) * (int64_t)sizeof(float);
      b_kstep = (TB ? (int64_t)BK * ldw : BK) * (int64_t)sizeof(float);
    }
    const int64_t kbase = GROUP ? 0 : (int64_t)split * nk * BK;  // first contraction index of this split
    const uint32_t ldx32 = (uint32_t)ldx, ldw32 = (uint32_t)ldw;  // launch_gemm keeps 256 rows x the pitch below 4 GB
    // contraction-major operands: a 16-B read is kept inside its row (those output rows / columns are never stored)
    const int cmax_a = M >= 4 ? ((M - 4) & ~3) : 0, cb_a = m0 < cmax_a ? m0 : cmax_a;
    const int cmax_b = N >= 4 ? ((N - 4) & ~3) : 0, cb_b = n0 < cmax_b ? n0 : cmax_b;
    base_a = reinterpret_cast<const char*>(TA ? X + kbase * ldx + cb_a : X + (int64_t)m0 * ldx + kbase);
    base_b = reinterpret_cast<const char*>(TB ? W + kbase * ldw + cb_b : W + (int64_t)n0 * ldw + kbase);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int vw = vw0 + u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (TA) {
          const int kr = vw * 4 + i;                 // k-row inside the step
          int col = m0 + lane * 4;
          col = col < cmax_a ? col : cmax_a;
          voa[u][i] = ((uint32_t)kr * ldx32 + (uint32_t)(col - cb_a)) * 4u;
        } else {
          const int r = vw * 32 + i * 8 + drow;      // row inside the A tile
          const int rr = r < M - m0 ? r : M - 1 - m0;  // ragged tiles re-read the last row; stores are guarded
          voa[u][i] = ((uint32_t)rr * ldx32 + (uint32_t)((dslot ^ ((r >> 1) & 7)) * 4)) * 4u;  // source-side swizzle
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (TB) {
          const int kr = vw * 4 + i * 2 + (lane >> 5);
          int col = n0 + (lane & 31) * 4;
          col = col < cmax_b ? col : cmax_b;
          vob[u][i] = ((uint32_t)kr * ldw32 + (uint32_t)(col - cb_b)) * 4u;
        } else {
          const int r = vw * 16 + i * 8 + drow;      // row inside the B tile
          const int rr = r < N - n0 ? r : N - 1 - n0;
          vob[u][i] = ((uint32_t)rr * ldw32 + (uint32_t)((dslot ^ ((r >> 1) & 7)) * 4)) * 4u;
        }
      }
    }
  };
  // One k-step's DMA = 6 instructions per wave, issued in three parts (A 0-1, A 2-3, B) so they can be spread
  // between the MFMA chunks of the previous step instead of queueing at the TA all at once.
  auto issue_dma_part = [&](int part, int kt, int stage, int u = 0) {
    const int vw = vw0 + u;
    float* sa = smem + stage * STAGE_FLOATS + (TA ? (vw * 4) * BM : (vw * 32) * BK);
    float* sb = smem + stage * STAGE_FLOATS + BM * BK + (TB ? (vw * 4) * BN : (vw * 16) * BK);
    if (part < 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = 2 * part + i;
        stlt_dma16(base_a + (int64_t)kt * a_kstep, voa[u][q], stlt_lds_addr(sa + q * (TA ? BM : 8 * BK)));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
        stlt_dma16(base_b + (int64_t)kt * b_kstep, vob[u][i], stlt_lds_addr(sb + i * (TB ? 2 * BN : 8 * BK)));
      }
    }
  };

  // ---- fragment side --------------------------------------------------------------------------------
  const int sw = (lr >> 1) & 7;
  const int a_row = (wm * 64 + lr) * BK;         // k-contiguous image; + 32*BK for the second M tile
  const int b_row = (BM + wn * 64 + lr) * BK;    // + 32*BK for the second N tile
  const int a_col2 = wm * 64 + 2 * lr;           // contraction-major image [32][BM]: this lane's pair of rows
  const int b_col2 = BM * BK + wn * 64 + 2 * lr; // contraction-major image [32][BN] behind the A image
  struct Frags { f32x4 a0, a1, b0, b1; };
  auto read_frags = [&](int stage, int c) {
    const float* s = smem + stage * STAGE_FLOATS;
    const int off = ((2 * c + lh) ^ sw) * 4;
    const int kk = 8 * c + 4 * lh;               // first of this lane's 4 contraction indices in the chunk
    Frags f;
    if (TA) {  // one ds_read_b64 = two adjacent m of one k: M-tile a holds rows wm*64 + 2*i + a (row-interleaved tiles)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(s + (kk + e) * BM + a_col2);
        f.a0[e] = t.x; f.a1[e] = t.y;
      }
    } else {
      f.a0 = *reinterpret_cast<const f32x4*>(s + a_row + off);
      f.a1 = *reinterpret_cast<const f32x4*>(s + a_row + 32 * BK + off);
    }
    if (TB) {  // N-tile b holds columns wn*64 + 2*j + b
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(s + (kk + e) * BN + b_col2);
        f.b0[e] = t.x; f.b1[e] = t.y;
      }
    } else {
      f.b0 = *reinterpret_cast<const f32x4*>(s + b_row + off);
      f.b1 = *reinterpret_cast<const f32x4*>(s + b_row + 32 * BK + off);
    }
    return f;
  };

  // The bias is the accumulators' initial value (one column per lane: n is fixed per lane and N-tile), so the
  // epilogue adds nothing.  The next tile's two bias values are fetched during the current tile's last k-step
  // It travels like the operands: wave 0 DMAs the tile's 128 bias values into a small LDS strip (double
  // buffered by tile parity) at the start of the previous tile's last k-step; that step's counted wait + barrier
  // publish it, and every wave then reads its two columns with ordinary ds_reads.  (A VGPR-destination load here
  // would make hipcc drain the DMA in flight with a vmcnt(0) at its first use.)
  float* bias_lds = smem + NSTAGE * STAGE_FLOATS;
  auto dma_bias = [&](int it) {
    if (bias && wave == (WS ? GEMM_WAVES : 0)) {
      int m0, n0, split;
      tile_origin(it, m0, n0, split);
      float* dst = bias_lds + (it & 1) * BN;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int n = n0 + i * 64 + lane;
        n = n < N ? n : N - 1;  // columns past N are never stored
        stlt_dma4(bias, (uint32_t)n * 4u, stlt_lds_addr(dst + i * 64));  // N < 2^30 here (bias only with the forward layout; launch_gemm checks): the byte offset fits 32 bits
      }
    }
  };
  if (WS && wave >= GEMM_WAVES) {
    // ---- loader waves: the whole DMA stream of the workgroup, two k-steps ahead of the MFMA waves, same barriers
    int l_it = 0, l_kt = kt_begin(0), l_stage = 0;
    bool l_fresh = true;  // row pointers not yet set for the tile the stream is in (first step, possibly mid-tile)
    auto l_step = [&]() {
      if ((STLT_GEMM_ABLATE & 32) ? l_fresh : (l_kt == 0 || l_fresh)) { dma_set_tile(l_it); l_fresh = false; }
      const int src_kt = (STLT_GEMM_ABLATE & 32) ? 0 : l_kt;
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        issue_dma_part(0, src_kt, l_stage, u);
        issue_dma_part(1, src_kt, l_stage, u);
        issue_dma_part(2, src_kt, l_stage, u);
      }
      if (++l_kt == nk) { ++l_it; l_kt = kt_begin(l_it); if (!(STLT_GEMM_ABLATE & 32)) l_fresh = true; }
      if (++l_stage == NSTAGE) l_stage = 0;
    };
    dma_bias(0);
    l_step();
    if (total_steps > 1) {
      l_step();
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // in-order counter: step 0 (and the bias strip before it) landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int w_it = 0, w_kt = kt_begin(0);  // position of the MFMA waves (the bias strip follows their tile changes)
    for (int step = 0; step < total_steps; ++step) {
      if (!GROUP && w_kt == nk - 1 && w_it + 1 < my_tiles) dma_bias(w_it + 1);
      if (!(STLT_GEMM_ABLATE & 1) && step + 2 < total_steps) {
        l_step();
        if (!(STLT_GEMM_ABLATE & 8)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // step+1 landed; only step+2's 12 instructions may stay in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (!GROUP && ++w_kt == nk) { ++w_it; w_kt = kt_begin(w_it); }
    }
    return;
  }
  if (!WS) dma_bias(0);
  f32x16 acc[2][2];
  // accumulator (a, b), register r of lane (lr, lh) holds output element
  //   row  = wm*64 + a*32 + lr                  (TA: wm*64 + 2*lr + a, the row-interleaved tiles of the ds_read_b64 fragments)
  //   col  = wn*64 + b*32 + i(r),  i(r) = (r&3) + 8*(r>>2) + 4*lh   (TB: wn*64 + 2*i(r) + b)
  // Initial value = the tile's bias (forward layout only): registers 4q..4q+3 are columns 8q+4lh .. +3 of the strip.
  auto init_acc = [&](int it, bool zero) {
    if (bias && !TB && !zero) {
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

  auto mfma_chunk = [&](const Frags& f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b0[e], f.a0[e], acc[0][0], 0, 0, 0);  // (W frag, X frag): D[n][m]
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b1[e], f.a0[e], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b0[e], f.a1[e], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b1[e], f.a1[e], acc[1][1], 0, 0, 0);
    }
  };

  // ---- prologue: two steps in flight ------------------------------------------------------------------
  int d_it = 0, d_kt = SK ? sk_kt0 : 0;  // DMA stream position (runs two steps ahead of the MFMAs)
  int d_stage = 0;
  if (!WS && SK && sk_kt0 != 0) dma_set_tile(0);  // a range that begins inside a tile
  auto dma_part = [&](int part) {  // part 0 also moves to the next tile's row pointers when needed
    if (WS) return;                // the loader waves own the DMA stream
    if (part == 0 && d_kt == 0) dma_set_tile(d_it);
    issue_dma_part(part, d_kt, d_stage);
    if (part == 2) {
      if (++d_kt == nk) { d_kt = 0; ++d_it; }
      if (++d_stage == NSTAGE) d_stage = 0;
    }
  };
  dma_part(0); dma_part(1); dma_part(2);
  if (WS) {
    // nothing of this wave's is in flight: the loaders' counted wait + this barrier publish step 0
  } else if (total_steps > 1) {
    dma_part(0); dma_part(1); dma_part(2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // in-order counter: step 0 (and the bias strip before it) landed
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  init_acc(0, SK && !seg_complete(0));
  Frags fa = read_frags(0, 0), fb;  // ping-pong fragment registers: 4 chunk reads per step, so fa is "current" at every step start

  int c_it = 0, c_kt = kt_begin(0);  // MFMA stream position
  if constexpr (GROUP) nk = grp.p[g_p0].nk;  // k-steps of the MFMA stream's current tile
  int stage = 0;
  constexpr bool RES_PF = STLT_GEMM_RES_PREFETCH && ADD && WS && !SK && !TA && !TB;
  f32x4 rpf[RES_PF ? 8 : 1];  // add-source pieces of row half a = 0 of the current tile, in flight during its last k-step
  bool rpf_ok = false;
  unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
#define GSTAMP(k) do { if (STAMP) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_now = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); t_acc[k] += t_now - t_prev; t_prev = t_now; __builtin_amdgcn_sched_barrier(0); } } while (0)
  if (STAMP) { t_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
  if (prio == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);  // static: the second-dispatched half wins arbitration
  for (int step = 0; step < total_steps; ++step) {
    if (prio == 2) { if (wave < 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    const int next_stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    const bool prefetch = (STLT_GEMM_ABLATE & 1) ? false : step + 2 < total_steps;  // step+2's operands go to the stage retired by the previous barrier
    const bool bias_step = c_kt == nk - 1 && c_it + 1 < my_tiles;
    if (!WS && bias_step) dma_bias(c_it + 1);  // older than this step's operand DMA: covered by the counted wait below
    if constexpr (RES_PF) {
      if (c_kt == nk - 1) {  // the epilogue is 64 MFMAs away: its first eight residual pieces travel under them
        int m0, n0, split;
        tile_origin(c_it, m0, n0, split);
        rpf_ok = (m0 + BM <= M) && (n0 + BN <= N) && (ldr & 3) == 0 && ((uintptr_t)R & 15) == 0;
        if (rpf_ok) {
          const float* rrow = R + (int64_t)(m0 + wm * 64 + lr) * ldr + n0;
#pragma unroll
          for (int h = 0; h < 8; ++h) rpf[h] = *reinterpret_cast<const f32x4*>(rrow + wn * 64 + (h >> 2) * 32 + 8 * (h & 3) + 4 * lh);
        }
      }
    }
    // chunks 0..2: read the next chunk of this stage, 16 MFMAs on the current one, a third of step+2's DMA
    fb = read_frags(stage, 1);
    mfma_chunk(fa);
    if (prefetch) dma_part(0);
    fa = read_frags(stage, 2);
    mfma_chunk(fb);
    if (prio == 2) { if (wave < 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
    if (prefetch) dma_part(1);
    fb = read_frags(stage, 3);
    mfma_chunk(fa);
    if (prefetch) dma_part(2);
    if (STAMP) asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][1][15]));
    GSTAMP(0);  // chunks 0..2: 48 MFMAs + fragment reads + DMA issue
    // chunk 3: retire this stage.  Every wave has received all its reads of `stage`, and step+1's DMA has
    // landed.  The VMEM counter is in order: the only operations younger than step+1's DMA that may stay in
    // flight are the 6 DMAs of step+2 (the bias strip and a previous epilogue's stores are older than those).
    if (WS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own fragment reads only; a previous epilogue's stores may stay in flight
    else if (prefetch) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    GSTAMP(1);  // wait for own DMA + LDS reads
    __builtin_amdgcn_s_barrier();
    GSTAMP(2);  // barrier
    GSTAMP(3);
    if (step + 1 < total_steps) fa = read_frags(next_stage, 0);
    mfma_chunk(fb);
    if (STAMP) asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][1][15]));
    GSTAMP(4);  // chunk 3: 16 MFMAs + next step's first fragment read
    stage = next_stage;

    if (++c_kt == nk || (SK && step == total_steps - 1)) {
      // ---- epilogue of tile c_it (the next tile's first fragments are already in registers, its next two
      // k-steps are in flight; the barrier of this last k-step published the next tile's bias strip).
      int m0, n0, split;
      tile_origin(c_it, m0, n0, split);
      if constexpr (GROUP) {  // the output side of this tile's product
        int pi, lt;
        group_tile(c_it, pi, lt);
        Y = grp.p[pi].c; ldy = grp.p[pi].ldc; R = grp.p[pi].r; ldr = grp.p[pi].ldr; M = grp.p[pi].M; N = grp.p[pi].N;
      }
      float* Yt = Y + (int64_t)split * slab_stride;
      const bool partial = SK && !seg_complete(c_it);
      // 16-byte stores need 16-byte aligned rows (wave-uniform test); otherwise, and on ragged tiles, guarded scalars
      const bool vec_ok = (m0 + BM <= M) && (n0 + BN <= N) && (ldy & 3) == 0 && ((uintptr_t)Yt & 15) == 0 &&
                          (!ADD || ((ldr & 3) == 0 && ((uintptr_t)R & 15) == 0));
      auto t_row = [&](int a) { return TA ? wm * 64 + 2 * lr + a : wm * 64 + a * 32 + lr; };
      // group g = 0..7 of an M tile a: 4 consecutive columns starting at t_col4(g), held in regs() of acc[a][*]
      //   !TB: g = 4*b + q        -> columns wn*64 + b*32 + 8q + 4lh + j      = acc[a][b][4q + j]
      //    TB: g = 2*q + half     -> columns wn*64 + 16q + 8lh + 4half + j    = acc[a][j&1][4q + 2half + (j>>1)]
      auto t_col4 = [&](int g) { return TB ? wn * 64 + 16 * (g >> 1) + 8 * lh + 4 * (g & 1) : wn * 64 + (g >> 2) * 32 + 8 * (g & 3) + 4 * lh; };
      auto group = [&](int a, int g) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v[j] = TB ? acc[a][j & 1][4 * (g >> 1) + 2 * (g & 1) + (j >> 1)] : acc[a][g >> 2][4 * (g & 3) + j];
        return v;
      };
      if (partial) {  // raw accumulators into this workgroup's head / tail slot (a whole BMxBN image, no guards)
        float* P = partials + (size_t)(2 * v + (c_it == hy_R ? 0 : 1)) * (BM * BN);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int g = 0; g < 8; ++g) *reinterpret_cast<f32x4*>(P + t_row(a) * BN + t_col4(g)) = group(a, g);
      } else if (ACT == STLT_ACT_GELU_BWD && vec_ok) {
        // du = drop(dh) ∘ gelu'(u) + the tile's column sums: the column groups go in batches of GB, a batch's u pieces
        // (2 rows x GB groups) requested together; a group's two rows are summed per lane, then over the 32 lanes of the half
        // wave (the wave's 64 rows); the wave's sums go to slot wm of the tile row, slots wm + 4 / 8 / 12 get zeros (the
        // fix-up kernel of a split tile fills all 16 slots itself)
        const uint64_t key = stlt_drop_key(epi.dr, epi.site);
        float* cs_row = epi.cs_part + ((size_t)(m0 / BM) * 16 + wm) * (size_t)N + n0;
        uint64_t drow[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) drow[a] = epi.drop_rows ? (uint64_t)epi.drop_rows[m0 + t_row(a)] : (uint64_t)(m0 + t_row(a));
        constexpr int GB = 2;  // column groups per batch of u loads (4 per batch spill 6-13 registers at the 168 budget)
#pragma unroll
        for (int hf = 0; hf < 8 / GB; ++hf) {
          f32x4 uv[2][GB];
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < GB; ++q) uv[a][q] = *reinterpret_cast<const f32x4*>(R + (int64_t)(m0 + t_row(a)) * ldr + n0 + t_col4(hf * GB + q));
#pragma unroll
          for (int q = 0; q < GB; ++q) {
            const int g = hf * GB + q;
            f32x4 cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              const f32x4 val = gelu_bwd4(group(a, g), uv[a][q], epi, key, drow[a] * (uint64_t)N + (uint64_t)(n0 + t_col4(g)));
              *reinterpret_cast<f32x4*>(Yt + (int64_t)(m0 + t_row(a)) * ldy + n0 + t_col4(g)) = val;
              cs += val;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = half_wave_sum(cs[j]);
            if (lr == 0) {
              *reinterpret_cast<f32x4*>(cs_row + t_col4(g)) = cs;
#pragma unroll
              for (int k = 1; k < 4; ++k) *reinterpret_cast<f32x4*>(cs_row + (size_t)(4 * k) * N + t_col4(g)) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
        }
      } else if (ACT == STLT_ACT_GELU_BWD) {
        // ragged tile: the same with guards
        const uint64_t key = stlt_drop_key(epi.dr, epi.site);
        float* cs_row = epi.cs_part + ((size_t)(m0 / BM) * 16 + wm) * (size_t)N;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          f32x4 cs = {0.f, 0.f, 0.f, 0.f};
          const int n = n0 + t_col4(g);
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int m = m0 + t_row(a);
            const f32x4 acc4 = group(a, g);
            if (m < M) {
              const uint64_t drow1 = epi.drop_rows ? (uint64_t)epi.drop_rows[m] : (uint64_t)m;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (n + j < N) {
                  float v = acc4[j];
                  if (epi.dr.thr) v = stlt_keep_k(epi.dr.thr, key, drow1 * (uint64_t)N + (uint64_t)(n + j)) ? v * epi.dr.scale : 0.f;
                  v *= gelu_grad_epilogue(R[(int64_t)m * ldr + n + j]);
                  Yt[(int64_t)m * ldy + n + j] = v;
                  cs[j] += v;
                }
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) cs[j] = half_wave_sum(cs[j]);
          if (lr == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (n + j < N) {
                cs_row[n + j] = cs[j];
#pragma unroll
                for (int k = 1; k < 4; ++k) cs_row[(size_t)(4 * k) * N + n + j] = 0.f;
              }
          }
        }
      } else if (vec_ok) {
        // whole tile, 16-byte aligned rows: straight-line code, 16 stores of 16 bytes per lane
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int m = m0 + t_row(a);
          float* yrow = Yt + (int64_t)m * ldy + n0;
          const float* rrow = ADD ? R + (int64_t)m * ldr + n0 : nullptr;
          // add-source (residual / residual gradient): the 8 pieces of this row pair are requested before its first store —
          // left inside the store loop, every load waits out its own latency behind the previous store (possible aliasing);
          // all 16 at once spill 24-44 of the 168 registers
          constexpr int RB = 8;
          f32x4 rv[ADD ? RB : 1];
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            if (ADD && g % RB == 0) {
              if (RES_PF && a == 0 && rpf_ok) {
#pragma unroll
                for (int h = 0; h < RB; ++h) rv[ADD ? h : 0] = rpf[RES_PF ? h : 0];
              } else {
#pragma unroll
                for (int h = 0; h < RB; ++h) rv[ADD ? h : 0] = *reinterpret_cast<const f32x4*>(rrow + t_col4(g + h));
              }
            }
            f32x4 val = group(a, g);
            if (ADD) val += rv[ADD ? g % RB : 0];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (ACT == STLT_ACT_GELU) val[j] = gelu_epilogue(val[j]);
              if (ACT == STLT_ACT_RELU) val[j] = fmaxf(val[j], 0.f);
            }
            if (STLT_GEMM_ABLATE & 16) asm volatile("" :: "v"(val));
            else if (STLT_GEMM_STORE_NT) __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(yrow + t_col4(g)));
            else *reinterpret_cast<f32x4*>(yrow + t_col4(g)) = val;
          }
        }
      } else {
        // ragged tile or unaligned rows: guarded scalars
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int m = m0 + t_row(a);
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const int n = n0 + t_col4(g);
            const f32x4 val = group(a, g);
            float* yp = Yt + (int64_t)m * ldy + n;
            const float* rp = ADD ? R + (int64_t)m * ldr + n : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float x = val[j];
              if (m < M && n + j < N) {
                if (ADD) x += rp[j];
                if (ACT == STLT_ACT_GELU) x = gelu_epilogue(x);
                if (ACT == STLT_ACT_RELU) x = fmaxf(x, 0.f);
                yp[j] = x;
              }
            }
          }
        }
      }
      if (c_it + 1 < my_tiles) init_acc(c_it + 1, SK && !seg_complete(c_it + 1));
      ++c_it;
      c_kt = kt_begin(c_it);
      if constexpr (GROUP) {
        if (c_it < my_tiles) { int pi, lt; group_tile(c_it, pi, lt); nk = grp.p[pi].nk; }
      }
      GSTAMP(5);  // epilogue
    }
  }
  if (dbg && tid == 0) {
    dbg[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    dbg[4 * (size_t)gridDim.x + (size_t)gridDim.x * GEMM_WAVES * 6 + 1024 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
  }
  if (STAMP && dbg && lane == 0) {
    unsigned long long* o = dbg + 4 * (size_t)gridDim.x + ((size_t)blockIdx.x * GEMM_WAVES + wave) * 6;
    for (int k = 0; k < 6; ++k) o[k] = t_acc[k];
  }
#undef GSTAMP
}


constexpr int FIXUP_CHUNKS = 16;
// Second half of a stream-K launch: FIXUP_CHUNKS workgroups per output tile.  A tile whose k-steps were all computed by one
// workgroup was stored by it; otherwise the segments' raw partial images are summed in workgroup order, then
// bias / add-source / activation are applied exactly as the main kernel's epilogue does.
template <int ACT>
__global__ __launch_bounds__(256) void gemm_fixup_kernel(const float* __restrict__ partials, int S, int nk,
                                                         const float* __restrict__ bias, const float* __restrict__ R,
                                                         int64_t ldr, float* __restrict__ Y, int64_t ldy, int M, int N,
                                                         int tiles_m, int tiles_n, int G, const SkPlan plan, const StltGemmEpi epi) {
  __shared__ float cs_red[8][BN];  // fused GELU backward: the column sums of this workgroup's rows
  const int t = blockIdx.x / FIXUP_CHUNKS, chunk = blockIdx.x - t * FIXUP_CHUNKS;  // a workgroup sums BM/FIXUP_CHUNKS rows of a tile
  const int lo = t * nk, hi = lo + nk;  // hybrid: t counts the tail tiles, steps are tail steps
  const bool hy = plan.dp_rounds > 0;
  const int gx = G >> 3;
  // workgroup (in range order) that owns k-step s, and the first step of workgroup v's range
  auto wg_of = [&](int s) {
    if (!hy) return s / S;
    int x = 0;
    while (x < 7 && s >= plan.P[x + 1]) ++x;
    return x * gx + (s - plan.P[x]) / plan.S[x];
  };
  auto start_of = [&](int v) {
    if (!hy) return v * S;
    const int x = v / gx;
    return plan.P[x] + (v - x * gx) * plan.S[x];
  };
  const int v_first = wg_of(lo), v_last = wg_of(hi - 1);
  if (v_first == v_last) return;
  int m0, n0;
  if (hy) {  // position of the tail tile in the band-major order of the main kernel
    constexpr int GROUP_M = 4;
    const int p = plan.dp_rounds * G + t;
    const int band = p / (GROUP_M * tiles_n), w = p - band * (GROUP_M * tiles_n);
    const int rows = tiles_m - band * GROUP_M < GROUP_M ? tiles_m - band * GROUP_M : GROUP_M;
    const int tn_g = w / rows;
    m0 = (band * GROUP_M + (w - tn_g * rows)) * BM;
    n0 = tn_g * BN;
  } else {
    const int tm = t / tiles_n;
    m0 = tm * BM;
    n0 = (t - tm * tiles_n) * BN;
  }
  const int c4 = (threadIdx.x & 31) * 4;
  float bv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bv[e] = (bias && n0 + c4 + e < N) ? bias[n0 + c4 + e] : 0.f;
  const int r_end = (chunk + 1) * (BM / FIXUP_CHUNKS);
  const uint64_t key = ACT == STLT_ACT_GELU_BWD ? stlt_drop_key(epi.dr, epi.site) : 0ull;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  for (int rr = chunk * (BM / FIXUP_CHUNKS) + (threadIdx.x >> 5); rr < r_end; rr += 8) {
    const int m = m0 + rr;
    if (m >= M) break;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int v = v_first; v <= v_last; ++v) {
      if (hy && start_of(v) >= plan.P[v / gx + 1]) continue;  // the last workgroups of an XCD may have no range
      const size_t slot = (size_t)(2 * v + (start_of(v) >= lo ? 0 : 1));
      acc += *reinterpret_cast<const f32x4*>(partials + slot * (BM * BN) + rr * BN + c4);
    }
    const uint64_t drow = (ACT == STLT_ACT_GELU_BWD && epi.drop_rows) ? (uint64_t)epi.drop_rows[m] : (uint64_t)m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + c4 + e;
      if (n < N) {
        float val = acc[e] + bv[e];
        if (ACT == STLT_ACT_GELU_BWD) {  // du = drop(dh) * gelu'(u), u read through the add-source pointer
          if (epi.dr.thr) val = stlt_keep_k(epi.dr.thr, key, drow * (uint64_t)N + (uint64_t)n) ? val * epi.dr.scale : 0.f;
          val *= gelu_grad_epilogue(R[(int64_t)m * ldr + n]);
          cs[e] += val;
        } else if (R) {
          val += R[(int64_t)m * ldr + n];
        }
        if (ACT == STLT_ACT_GELU) val = gelu_epilogue(val);
        if (ACT == STLT_ACT_RELU) val = fmaxf(val, 0.f);
        Y[(int64_t)m * ldy + n] = val;
      }
    }
  }
  if (ACT == STLT_ACT_GELU_BWD) {  // slot `chunk` of the tile row: this workgroup's rows, summed in row-thread order
#pragma unroll
    for (int e = 0; e < 4; ++e) cs_red[threadIdx.x >> 5][c4 + e] = cs[e];
    __syncthreads();
    if (threadIdx.x < BN && n0 + (int)threadIdx.x < N) {
      float tsum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) tsum += cs_red[k][threadIdx.x];
      epi.cs_part[((size_t)(m0 / BM) * 16 + chunk) * (size_t)N + n0 + threadIdx.x] = tsum;
    }
  }
}

// Fix-up of a grouped stream-K launch (S k-steps per workgroup over the group's flattened k-step space): tile t of the
// launch order belongs to product p = the last one with tile_base[p] <= t; its segments are summed in workgroup order and
// added to the product's add-source (the weight gradient being accumulated).
__global__ __launch_bounds__(256) void gemm_fixup_group_kernel(const float* __restrict__ partials, int S, const StltGemmGroup grp) {
  const int t = blockIdx.x / FIXUP_CHUNKS, chunk = blockIdx.x - t * FIXUP_CHUNKS;
  int pi = 0;
  while (grp.tile_base[pi + 1] <= t) ++pi;
  const int lt = t - grp.tile_base[pi];
  const int nk = grp.p[pi].nk;
  const int lo = grp.step_base[pi] + lt * nk, hi = lo + nk;
  const int v_first = lo / S, v_last = (hi - 1) / S;
  if (v_first == v_last) return;  // one workgroup computed the whole tile and stored it
  const int tn_all = grp.p[pi].tiles_n;
  const int tm = lt / tn_all;
  const int m0 = tm * BM, n0 = (lt - tm * tn_all) * BN;
  const int M = grp.p[pi].M, N = grp.p[pi].N;
  const float* __restrict__ R = grp.p[pi].r;
  float* __restrict__ Y = grp.p[pi].c;
  const int64_t ldr = grp.p[pi].ldr, ldy = grp.p[pi].ldc;
  const int c4 = (threadIdx.x & 31) * 4;
  const int r_end = (chunk + 1) * (BM / FIXUP_CHUNKS);
  for (int rr = chunk * (BM / FIXUP_CHUNKS) + (threadIdx.x >> 5); rr < r_end; rr += 8) {
    const int m = m0 + rr;
    if (m >= M) break;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int v = v_first; v <= v_last; ++v) {
      const size_t slot = (size_t)(2 * v + (v * S >= lo ? 0 : 1));
      acc += *reinterpret_cast<const f32x4*>(partials + slot * (BM * BN) + rr * BN + c4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + c4 + e;
      if (n < N) {
        float val = acc[e];
        if (R) val += R[(int64_t)m * ldr + n];
        Y[(int64_t)m * ldy + n] = val;
      }
    }
  }
}

thread_local float* t_gemm_scratch = nullptr;
thread_local size_t t_gemm_scratch_bytes = 0;

// dst[i] = (accumulate ? dst[i] : 0) + sum_s slabs[s*stride + i] : the deterministic second half of a split-K product
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int64_t stride, int n_slabs,
                                                           float* __restrict__ dst, int64_t n, int accumulate) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      f32x4 acc = accumulate ? *reinterpret_cast<const f32x4*>(dst + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < n_slabs; ++s) acc += *reinterpret_cast<const f32x4*>(slabs + s * stride + i);
      *reinterpret_cast<f32x4*>(dst + i) = acc;
    } else {
      for (int64_t j = i; j < n; ++j) {
        float acc = accumulate ? dst[j] : 0.f;
        for (int s = 0; s < n_slabs; ++s) acc += slabs[s * stride + j];
        dst[j] = acc;
      }
    }
  }
}

// Many partial rows, few columns (LayerNorm dw/db and bias-gradient partials): a block owns 16 columns, its 16
// "slab lanes" each sum every 16th partial row, then a fixed-order LDS pass adds the 16 lanes: deterministic.
__global__ __launch_bounds__(256) void reduce_tall_kernel(const float* __restrict__ slabs, int64_t stride, int n_slabs,
                                                          float* __restrict__ dst, int64_t n, int accumulate) {
  __shared__ float part[16][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t j = (int64_t)blockIdx.x * 16 + col;
  float acc = 0.f;
  if (j < n)
    for (int s = sl; s < n_slabs; s += 16) acc += slabs[s * stride + j];
  part[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && j < n) {
    float t = accumulate ? dst[j] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][col];
    dst[j] = t;
  }
}

// The same reduction for up to three destinations at once: column j of the slabs belongs to destination j / n (LayerNorm
// backward leaves [dw | db | column sums] per partial row): one launch instead of three, same summation order.
__global__ __launch_bounds__(256) void reduce_tall3_kernel(const float* __restrict__ slabs, int64_t stride, int n_slabs,
                                                           float* __restrict__ dst0, float* __restrict__ dst1,
                                                           float* __restrict__ dst2, int64_t n, int accumulate) {
  __shared__ float part[16][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t j = (int64_t)blockIdx.x * 16 + col;  // n is a multiple of 16 here, so a block never straddles destinations
  const int which = (int)(j / n);
  float* dst = which == 0 ? dst0 : which == 1 ? dst1 : dst2;
  const bool live = j < 3 * n && dst != nullptr;
  float acc = 0.f;
  if (live)
    for (int s = sl; s < n_slabs; s += 16) acc += slabs[s * stride + j];
  part[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && live) {
    const int64_t c = j - (int64_t)which * n;
    float t = accumulate ? dst[c] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][col];
    dst[c] = t;
  }
}

// Batched form of the two kernels above: block b belongs to entry e with blk_base[e] <= b < blk_base[e + 1]; per destination column the
// same summation order (16 interleaved partial sums over the slabs, then the 16 parts in order, then += dst).
struct ReduceBatch { StltReduceEntry e[STLT_REDUCE_DEFER_MAX]; int blk_base[STLT_REDUCE_DEFER_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void reduce_batch_kernel(const ReduceBatch b) {
  __shared__ float part[16][17];
  int ei = 0;
  while (ei + 1 < b.n && (int)blockIdx.x >= b.blk_base[ei + 1]) ++ei;
  const StltReduceEntry& en = b.e[ei];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t j = (int64_t)((int)blockIdx.x - b.blk_base[ei]) * 16 + col;
  const int which = en.n_dst == 1 ? 0 : (int)(j / en.n);  // n % 16 == 0 for multi-destination entries: a block never straddles destinations
  float* dst = which < en.n_dst ? (which == 0 ? en.dst[0] : which == 1 ? en.dst[1] : en.dst[2]) : nullptr;
  const bool live = j < (int64_t)en.n_dst * en.n && dst != nullptr;
  float acc = 0.f;
  if (live)
    for (int s = sl; s < en.n_slabs; s += 16) acc += en.slabs[s * en.stride + j];
  part[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && live) {
    const int64_t c = j - (int64_t)which * en.n;
    float t = dst[c];
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][col];
    dst[c] = t;
  }
}

thread_local StltReduceDefer* t_reduce_defer = nullptr;

}  // namespace

void stlt_reduce_defer_set(StltReduceDefer* d) { t_reduce_defer = d; }

int stlt_reduce_defer_flush(StltReduceDefer* d) {
  if (!d) return 0;
  if (d->n == 0) { const int e = d->err; d->err = 0; return e; }
  ReduceBatch b;
  int blocks = 0;
  for (int i = 0; i < d->n; ++i) {
    b.e[i] = d->e[i];
    b.blk_base[i] = blocks;
    blocks += (int)(((int64_t)d->e[i].n_dst * d->e[i].n + 15) / 16);
  }
  for (int i = d->n; i <= STLT_REDUCE_DEFER_MAX; ++i) b.blk_base[i] = blocks;
  b.n = d->n;
  d->n = 0;
  StltProfScope ps(STLT_K_MISC, d->s);
  hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, d->s, b);
  int e = stlt_check_launch("reduce_batch_kernel");
  if (!e) { e = d->err; d->err = 0; }
  return e;
}

// next chunk of the pool for a producer's partial rows; a full pool first flushes the entries that still read it
float* stlt_reduce_defer_chunk(StltReduceDefer* d, size_t floats, float* fallback, int* err) {
  *err = 0;
  if (!d || !d->pool || floats > d->pool_floats) return fallback;
  floats = (floats + 63) / 64 * 64;
  if (d->used + floats > d->pool_floats) {
    *err = stlt_reduce_defer_flush(d);
    if (*err && !d->err) d->err = *err;  // callers that cannot return it (pointer-valued helpers) see it at their final flush
    d->used = 0;
  }
  float* p = d->pool + d->used;
  d->used += floats;
  return p;
}

static bool reduce_defer_take(const float* slabs, int64_t stride, int n_slabs, float* d0, float* d1, float* d2, int n_dst, int64_t n, hipStream_t s, int* err) {
  StltReduceDefer* d = t_reduce_defer;
  *err = 0;
  if (!d || s != d->s || !d->pool || slabs < d->pool || slabs >= d->pool + d->pool_floats) return false;  // only partial rows that live in the pool
  if (((int64_t)n_dst * n + 15) / 16 > 0x3fffffLL) return false;
  bool clash = d->n == STLT_REDUCE_DEFER_MAX;  // a destination may appear once per batch (its blocks read-modify-write it)
  for (int i = 0; i < d->n && !clash; ++i)
    for (int a = 0; a < d->e[i].n_dst; ++a) {
      float* q = d->e[i].dst[a];
      if (q && (q == d0 || (n_dst > 1 && (q == d1 || q == d2)))) clash = true;
    }
  if (clash) { *err = stlt_reduce_defer_flush(d); if (*err) return true; }
  StltReduceEntry& e = d->e[d->n++];
  e.slabs = slabs; e.dst[0] = d0; e.dst[1] = d1; e.dst[2] = d2; e.stride = stride; e.n = n; e.n_slabs = n_slabs; e.n_dst = n_dst;
  return true;
}

// workgroups a launch may use: the device's CUs, or fewer while a StltGemmWgCap is alive on the calling thread (two persistent
// launches on two streams can only share the chip if their grids add up to the CU count: one 144-KB workgroup fits a CU)
thread_local int t_gemm_wg_cap = 0;
static int n_cu() {
  const int n = stlt_device_cus();
  return (t_gemm_wg_cap > 0 && t_gemm_wg_cap < n) ? t_gemm_wg_cap : n;
}

// Relative XCD speeds for the hybrid launch's tail ranges: STLT_GEMM_XCD_W="w0,...,w7" (experiments) — empty: equal.
static const double* xcd_weights() {
  static double w[8];
  static const bool have = [] {
    const char* e = getenv("STLT_GEMM_XCD_W");
    if (!e) return false;
    int n = 0;
    for (const char* p = e; *p && n < 8;) {
      char* end = nullptr;
      const double v = strtod(p, &end);
      if (end == p) break;
      w[n++] = v;
      p = *end == ',' ? end + 1 : end;
    }
    if (n != 8) return false;
    for (int i = 0; i < 8; ++i) if (!(w[i] > 0.5 && w[i] < 2.0)) return false;
    return true;
  }();
  return have ? w : nullptr;
}

// Hybrid plan for n_tiles tiles of nk k-steps on G = 8*gx workgroups: R whole-tile rounds, then the remaining tiles as
// k-step ranges; with XCD weights the ranges are sized so that every XCD finishes its rounds + range at the same time.
// Returns false when the shape leaves no room (the caller keeps the plain assignment).
static bool make_sk_plan(int64_t n_tiles, int64_t nk, int64_t G, const double* w, bool force_tail_round, SkPlan& plan, int64_t& tail_tiles) {
  if ((G & 7) != 0 || n_tiles <= G) return false;
  const int64_t gx = G >> 3;
  int64_t R = n_tiles / G, tail = n_tiles - R * G;
  if (tail * nk < 4 * G || (force_tail_round && tail == 0)) { R -= 1; tail += G; }
  const int64_t total = n_tiles * nk;
  double want[8], sum = 0.0;
  for (;;) {  // weighted: move whole rounds into the tail until the slowest XCD's share of it is not negative
    if (R < 1 || tail <= 0) return false;
    bool ok = true;
    sum = 0.0;
    for (int x = 0; x < 8; ++x) {
      double sx = (double)(tail * nk) / 8.0;  // tail steps of XCD x (all of its workgroups)
      if (w) {
        double sw = 0.0;
        for (int i = 0; i < 8; ++i) sw += w[i];
        sx = w[x] * (double)total / sw - (double)(R * nk * gx);  // its share of everything, minus its whole-tile rounds
        if (sx < 0.25 * (double)(nk * gx)) ok = false;
      }
      want[x] = sx;
      sum += sx;
    }
    if (ok) break;
    R -= 1;
    tail += G;
  }
  const int64_t tail_steps = tail * nk;
  if (!(sum > 0.0)) return false;
  plan.dp_rounds = (int)R;
  plan.P[0] = 0;
  double cum = 0.0;
  for (int x = 0; x < 8; ++x) {
    cum += want[x];
    int64_t p = x == 7 ? tail_steps : (int64_t)((double)tail_steps * (cum / sum) + 0.5);
    if (p < plan.P[x]) p = plan.P[x];
    if (p > tail_steps) p = tail_steps;
    plan.P[x + 1] = (int)p;
    const int64_t len = p - plan.P[x];
    plan.S[x] = (int)((len + gx - 1) / gx);
    if (plan.S[x] < 1) plan.S[x] = 1;
  }
  tail_tiles = tail;
  return true;
}

static int launch_gemm_impl(int transA, int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                            const float* r, int64_t ldr, float* c, int64_t ldc, int64_t slab_stride, int64_t M, int64_t N,
                            int64_t K, int n_split, int act, hipStream_t s, const StltGemmEpi* epi_in, bool on_copy);

// C (M,N) = opA(A)·opB(B) [+ bias | + R], contraction length K (multiple of 32).  n_split > 1: C is a slab buffer.
int launch_gemm(int transA, int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                const float* r, int64_t ldr, float* c, int64_t ldc, int64_t slab_stride, int64_t M, int64_t N,
                int64_t K, int n_split, int act, hipStream_t s, const StltGemmEpi* epi_in) {
  // An input gradient dX (M, N = k_in) = dY (M, K = n_out) · W (n_out, k_in) whose weight has a current transposed copy in the call's
  // training context (wt_cache.hip: the trainer refreshed it at the start of its step): the same product is the FORWARD layout on the
  // copy, dX = dY·(Wt)ᵀ — operand fragments by 16-byte LDS reads instead of the NN layout's four-byte gather of W (0.84 - 0.89 of the
  // peak against the forward's 0.888 on the 14 336-row spatial products), add-source and fused GELU backward included.
  if (!transA && n_split == 1 && M > 0 && M <= 128 && a && b && c && !(epi_in && act == STLT_ACT_GELU_BWD)) {  // a few rows: split-k partial tiles (gemm_any.hip)
    bool taken = false;
    if (int e = launch_gemm_skinny(transB, a, lda, b, ldb, bias, r, ldr, c, ldc, M, N, K, act, s, &taken)) return e;
    if (taken) return 0;
  }
  // A weight gradient over a few rows (dW (n_out, k_in) += dYᵀ·X with at most 128 rows to contract: the heads and the one-row-per-clip tail of a
  // 64-clip step): 2 - 4 k-slabs per 256 x 128 tile leave the large kernel a launch of partial tiles + a fix-up (43 us for 768 x 768 x 64);
  // 64 x 64 whole tiles on the compatibility kernel fill the chip with one launch
  static const bool skinny_dw = [] { const char* e = getenv("STLT_GEMM_SKINNY"); return !(e && e[0] == '0'); }();
  if (skinny_dw && transA && transB && n_split == 1 && K > 0 && K <= 128 && !bias && act == STLT_ACT_NONE && a && b && c && M > 0 && (!r || ldr >= N))
    return launch_gemm_any(1, 1, a, lda, b, ldb, nullptr, r, ldr, c, ldc, M, N, K, STLT_ACT_NONE, s);
  static const bool copy_on = [] { const char* e = getenv("STLT_GEMM_DX_WT"); return !(e && e[0] == '0'); }();  // A/B knob
  if (copy_on && !transA && transB && n_split == 1 && b && ldb == N && M > 0) {
    const float* wt = nullptr;
    int64_t ldwt = 0;
    if (stlt_wt_lookup(b, K, N, &wt, &ldwt)) {
      const int rc = launch_gemm_impl(0, 0, a, lda, wt, ldwt, bias, r, ldr, c, ldc, slab_stride, M, N, K, n_split, act, s, epi_in, true);
      if (rc == 0) stlt_wt_count_hit();
      return rc;
    }
  }
  return launch_gemm_impl(transA, transB, a, lda, b, ldb, bias, r, ldr, c, ldc, slab_stride, M, N, K, n_split, act, s, epi_in, false);
}

static int launch_gemm_impl(int transA, int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                            const float* r, int64_t ldr, float* c, int64_t ldc, int64_t slab_stride, int64_t M, int64_t N,
                            int64_t K, int n_split, int act, hipStream_t s, const StltGemmEpi* epi_in, bool on_copy) {
  if (!a || !b || !c) return stlt_set_error(STLT_EINVAL, "gemm: null pointer");
  const StltGemmEpi epi = epi_in ? *epi_in : StltGemmEpi{StltDrop{0u, 1.0f, 0ull}, 0u, nullptr, nullptr};
  if (act == STLT_ACT_GELU_BWD) {
    if (!epi_in || !epi.cs_part || !r || transA || (!transB && !on_copy) || n_split != 1 || bias)
      return stlt_set_error(STLT_EINVAL, "gemm: the fused GELU backward is the dX layout with u as the add-source and a column-sum buffer");
  }
  if (M < 0 || N <= 0 || K <= 0) return stlt_set_error(STLT_EINVAL, "gemm: bad shape (M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
  if (K % BK != 0 || lda % 4 != 0 || ldb % 4 != 0) {  // a partial k-slab / rows that are not 16-byte aligned cannot be staged by LDS-DMA:
    // the fallback kernel of gemm_any.hip (hidden sizes that are not multiples of 32)
    if (n_split != 1 || act == STLT_ACT_GELU_BWD)
      return stlt_set_error(STLT_EINVAL, "gemm: a split product / the fused GELU backward needs K=%lld to be a multiple of %d and pitches that are multiples of 4", (long long)K, BK);
    return launch_gemm_any(transA, transB, a, lda, b, ldb, bias, r, ldr, c, ldc, M, N, K, act, s);
  }
  if (lda % 4 != 0 || ldb % 4 != 0 || (r && ldr < N) || ldc < N || lda < (transA ? M : K) || ldb < (transB ? N : K))
    return stlt_set_error(STLT_EINVAL, "gemm: bad leading dimension (lda=%lld ldb=%lld ldc=%lld)", (long long)lda, (long long)ldb, (long long)ldc);
  if (M > 0x7fffff00LL || N > 0x7fffff00LL || (bias && N > 0x3fffff00LL)) return stlt_set_error(STLT_EINVAL, "gemm: M/N too large");
  if (lda > 0x3fffffLL || ldb > 0x3fffffLL)  // the DMA addresses a tile's 256 rows with 32-bit byte offsets from the tile's origin
    return stlt_set_error(STLT_EINVAL, "gemm: row pitch too large (lda=%lld ldb=%lld, at most 4194303 floats)", (long long)lda, (long long)ldb);
  if (act != STLT_ACT_NONE && act != STLT_ACT_GELU && act != STLT_ACT_RELU && act != STLT_ACT_GELU_BWD) return stlt_set_error(STLT_EINVAL, "gemm: unknown activation %d", act);
  if (n_split < 1 || (K / BK) % n_split != 0) return stlt_set_error(STLT_EINVAL, "gemm: n_split=%d must divide K/32=%lld", n_split, (long long)(K / BK));
  if (transA && !transB) return stlt_set_error(STLT_EINVAL, "gemm: the (transA, !transB) layout is not built");
  if (n_split > 1 && (bias || act != STLT_ACT_NONE || r)) return stlt_set_error(STLT_EINVAL, "gemm: a split product takes no bias / activation / add-source");
  if ((transA || transB) && ((act != STLT_ACT_NONE && act != STLT_ACT_GELU_BWD) || bias)) return stlt_set_error(STLT_EINVAL, "gemm: bias/activation only with the forward (NT) layout");
  if (r && act != STLT_ACT_NONE && act != STLT_ACT_GELU_BWD) return stlt_set_error(STLT_EINVAL, "gemm: an add-source excludes an activation");
  if (M == 0) return 0;
  const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  if (tiles_m * tiles_n * n_split > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "gemm: too many tiles");
  // persistent grid: one 144-KB workgroup per CU; no inter-workgroup dependency, so residency is only a speed matter
  int64_t n_wg = tiles_m * tiles_n * n_split;
  if (n_wg > n_cu()) n_wg = n_cu();
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)M * (double)N * (double)K);
  stlt_prof_note("gemm%s M=%lld N=%lld K=%lld act=%d%s tile=256x128 tiles=%lld", transA ? "(dW)" : (transB ? "(dX)" : (on_copy ? "(dX on the Wt copy)" : "")), (long long)M, (long long)N, (long long)K, act,
                 r ? "+R" : "", (long long)(tiles_m * tiles_n));
  dim3 block(GEMM_THREADS);
  // Stream-K when whole tiles would leave CUs idle (fewer tiles than CUs, or a ragged last round) and the caller
  // lent scratch for the partial tiles (StltGemmScratch / stlt_gemm_set_scratch).
  const int64_t n_tiles = tiles_m * tiles_n, nk = K / BK;
  const int64_t rounds = (n_tiles + n_cu() - 1) / n_cu();
  const double fill = (double)n_tiles / (double)(rounds * n_cu());
  static const double sk_fill = [] { const char* e = getenv("STLT_GEMM_SK_FILL"); return e ? atof(e) : 0.9; }();  // A/B knob
  // with XCD weights, well-filled launches of at least four rounds get a weighted tail too (if a plan exists)
  bool weighted = false;
  if (xcd_weights() && n_split == 1 && t_gemm_scratch && fill >= sk_fill && n_tiles >= 4 * (int64_t)n_cu() && n_tiles * nk < 0x7fffffffLL) {
    SkPlan probe{};
    int64_t tail = 0;
    weighted = make_sk_plan(n_tiles, nk, n_cu() < STLT_GEMM_SK_MAX_WG ? n_cu() : STLT_GEMM_SK_MAX_WG, xcd_weights(), true, probe, tail);
  }
  if (n_split == 1 && t_gemm_scratch && (fill < sk_fill || weighted) && n_tiles * nk < 0x7fffffffLL && !g_stlt_debug_buf) {
    int64_t G = n_cu() < STLT_GEMM_SK_MAX_WG ? n_cu() : STLT_GEMM_SK_MAX_WG;
    if (n_tiles * nk < 4 * G) G = (n_tiles * nk + 3) / 4;  // at least ~4 k-steps per workgroup
    if (t_gemm_scratch_bytes >= (size_t)(2 * G) * BM * BN * sizeof(float)) {
      const int S = (int)((n_tiles * nk + G - 1) / G);
      dim3 grid((unsigned)G);
      float* P = t_gemm_scratch;
      stlt_prof_note("stream-K wg=%lld ksteps/wg=%d (+fix-up)", (long long)G, S);
      static const bool ws_sk = [] { const char* e = getenv("STLT_GEMM_WS"); return e ? atoi(e) != 0 : (STLT_GEMM_WS_DEFAULT != 0); }();
      // hybrid: whole-tile rounds in the grouped order + a stream-K tail (loader-wave build, full grid of 8 x Gx workgroups,
      // the tail at least 4 k-steps per workgroup: the last whole round joins it otherwise).  STLT_GEMM_HYBRID=0: plain stream-K.
      static const bool hybrid_on = [] { const char* e = getenv("STLT_GEMM_HYBRID"); return e ? atoi(e) != 0 : true; }();
      SkPlan plan{};
      int64_t fix_tiles = n_tiles;
      if (hybrid_on && ws_sk && G == n_cu()) {
        int64_t tail = 0;
        if (make_sk_plan(n_tiles, nk, G, xcd_weights(), weighted, plan, tail)) fix_tiles = tail;
        else plan = SkPlan{};
      }
#define LAUNCH_SK1(ACTV, TAV, TBV, ADDV, WSV, BLK) hipLaunchKernelGGL((gemm_nt_kernel<ACTV, false, TAV, TBV, ADDV, true, WSV>), grid, BLK, 0, s, a, lda, b, ldb, bias, r, ldr, c, ldc, slab_stride, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n, 1, P, nullptr, plan, SkNone{}, karg<(ACTV) == STLT_ACT_GELU_BWD>(epi))
#define LAUNCH_SK(ACTV, TAV, TBV, ADDV) do { if (ws_sk) LAUNCH_SK1(ACTV, TAV, TBV, ADDV, true, dim3(GEMM_THREADS_WS)); else LAUNCH_SK1(ACTV, TAV, TBV, ADDV, false, block); } while (0)
      if (transA) { if (r) LAUNCH_SK(STLT_ACT_NONE, true, true, true); else LAUNCH_SK(STLT_ACT_NONE, true, true, false); }
      else if (transB && act == STLT_ACT_GELU_BWD) LAUNCH_SK(STLT_ACT_GELU_BWD, false, true, true);
      else if (transB) { if (r) LAUNCH_SK(STLT_ACT_NONE, false, true, true); else LAUNCH_SK(STLT_ACT_NONE, false, true, false); }
      else if (act == STLT_ACT_GELU_BWD) LAUNCH_SK(STLT_ACT_GELU_BWD, false, false, true);  // the FFN hidden gradient on the transposed copy of W2
      else if (r) LAUNCH_SK(STLT_ACT_NONE, false, false, true);  // y = x·Wᵀ + b + r (the residual of a post-norm layer)
      else if (act == STLT_ACT_GELU) LAUNCH_SK(STLT_ACT_GELU, false, false, false);
      else if (act == STLT_ACT_RELU) LAUNCH_SK(STLT_ACT_RELU, false, false, false);
      else LAUNCH_SK(STLT_ACT_NONE, false, false, false);
#undef LAUNCH_SK
#undef LAUNCH_SK1
      if (int e = stlt_check_launch("gemm_nt_kernel(stream-k)")) return e;
      dim3 fgrid((unsigned)(fix_tiles * FIXUP_CHUNKS)), fblock(256);
#define FIX(ACTV) hipLaunchKernelGGL((gemm_fixup_kernel<ACTV>), fgrid, fblock, 0, s, P, S, (int)nk, bias, r, ldr, c, ldc, (int)M, (int)N, (int)tiles_m, (int)tiles_n, (int)G, plan, epi)
      if (act == STLT_ACT_GELU_BWD) FIX(STLT_ACT_GELU_BWD);
      else if (act == STLT_ACT_GELU) FIX(STLT_ACT_GELU);
      else if (act == STLT_ACT_RELU) FIX(STLT_ACT_RELU);
      else FIX(STLT_ACT_NONE);
#undef FIX
      return stlt_check_launch("gemm_fixup_kernel");
    }
  }
  dim3 grid((unsigned)n_wg);
  stlt_prof_note("wg=%lld rounds=%lld ksteps=%lld split=%d", (long long)n_wg, (long long)((tiles_m * tiles_n * n_split + n_wg - 1) / n_wg), (long long)(nk / n_split), n_split);
  // wave-specialised build (4 DMA-only waves beside the 8 MFMA waves) unless STLT_GEMM_WS=0 (A/B measurements)
  static const bool ws = [] { const char* e = getenv("STLT_GEMM_WS"); return e ? atoi(e) != 0 : (STLT_GEMM_WS_DEFAULT != 0); }();
  const dim3 block_ws(GEMM_THREADS_WS);
#define LAUNCH1(ACTV, STAMPV, TAV, TBV, ADDV, WSV, BLK) hipLaunchKernelGGL((gemm_nt_kernel<ACTV, STAMPV, TAV, TBV, ADDV, false, WSV>), grid, BLK, 0, s, a, lda, b, ldb, bias, r, ldr, c, ldc, slab_stride, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n, n_split, (float*)nullptr, g_stlt_debug_buf, SkNone{}, SkNone{}, karg<(ACTV) == STLT_ACT_GELU_BWD>(epi))
#define LAUNCH(ACTV, STAMPV, TAV, TBV, ADDV) do { if (ws) LAUNCH1(ACTV, STAMPV, TAV, TBV, ADDV, true, block_ws); else LAUNCH1(ACTV, STAMPV, TAV, TBV, ADDV, false, block); } while (0)
  if (transA) { if (r) LAUNCH(STLT_ACT_NONE, false, true, true, true); else LAUNCH(STLT_ACT_NONE, false, true, true, false); }
  else if (transB && act == STLT_ACT_GELU_BWD) LAUNCH(STLT_ACT_GELU_BWD, false, false, true, true);
  else if (transB) { if (r) LAUNCH(STLT_ACT_NONE, false, false, true, true); else LAUNCH(STLT_ACT_NONE, false, false, true, false); }
  else if (act == STLT_ACT_GELU_BWD) LAUNCH(STLT_ACT_GELU_BWD, false, false, false, true);  // the FFN hidden gradient on the transposed copy of W2
  else if (r && g_stlt_debug_buf && getenv("STLT_GEMM_STAMP")) LAUNCH(STLT_ACT_NONE, true, false, false, true);  // diagnostic build path only (residual-add epilogue)
  else if (r) LAUNCH(STLT_ACT_NONE, false, false, false, true);  // y = x·Wᵀ + b + r
  else if (g_stlt_debug_buf && getenv("STLT_GEMM_STAMP")) LAUNCH(STLT_ACT_NONE, true, false, false, false);  // diagnostic build path only
  else if (act == STLT_ACT_GELU) LAUNCH(STLT_ACT_GELU, false, false, false, false);
  else if (act == STLT_ACT_RELU) LAUNCH(STLT_ACT_RELU, false, false, false, false);
  else LAUNCH(STLT_ACT_NONE, false, false, false, false);
#undef LAUNCH
#undef LAUNCH1
  return stlt_check_launch("gemm_nt_kernel");
}

bool stlt_gemm_has_scratch() { return t_gemm_scratch != nullptr && t_gemm_scratch_bytes >= STLT_GEMM_SCRATCH_BYTES; }
// the calling thread's lent scratch (stream-K partial tiles; between launches free for other stream-ordered uses such as the
// split-bf16 kernel's weight planes); nullptr when nothing is lent
float* stlt_gemm_scratch_ptr(size_t* bytes) { *bytes = t_gemm_scratch ? t_gemm_scratch_bytes : 0; return t_gemm_scratch; }

// g_w_i (n_out_i, k_in_i) += dy_i[:rows_i]ᵀ · x_i[:rows_i] for every item, as ONE stream-K launch + one fix-up: every CU gets
// an equal share of the group's k-steps, a workgroup's range may cross from one product into the next.  Deterministic
// (fixed summation order), like the per-product launches it replaces.
int launch_weight_grad_group(const StltWeightGradItem* items, int n_items, hipStream_t s) {
  if (!items || n_items < 1 || n_items > STLT_GEMM_GROUP_MAX) return stlt_set_error(STLT_EINVAL, "weight_grad_group: 1..%d items", STLT_GEMM_GROUP_MAX);
  if (!stlt_gemm_has_scratch()) return stlt_set_error(STLT_EWORKSPACE, "weight_grad_group: no stream-K scratch lent (StltGemmScratch)");
  StltGemmGroup grp{};
  int64_t tiles = 0, steps = 0;
  double flops = 0.0;
  int n = 0;
  for (int i = 0; i < n_items; ++i) {
    const StltWeightGradItem& it = items[i];
    if (!it.g_w || it.rows == 0) continue;
    if (!it.dy || !it.x || it.n_out <= 0 || it.k_in <= 0 || it.rows < 0 || it.rows % BK != 0 || it.n_out % 4 != 0 || it.k_in % 4 != 0)
      return stlt_set_error(STLT_EINVAL, "weight_grad_group: item %d: rows=%lld must be a multiple of %d, n_out=%lld / k_in=%lld multiples of 4", i,
                            (long long)it.rows, BK, (long long)it.n_out, (long long)it.k_in);
    if (it.n_out > 0x3fffffLL || it.k_in > 0x3fffffLL || it.rows > 0x7fffff00LL) return stlt_set_error(STLT_EINVAL, "weight_grad_group: item too large");
    const int64_t tm = (it.n_out + BM - 1) / BM, tn = (it.k_in + BN - 1) / BN;
    StltGemmProblem& q = grp.p[n];
    q.a = it.dy; q.lda = (int)it.n_out; q.b = it.x; q.ldb = (int)it.k_in; q.r = it.g_w; q.ldr = (int)it.k_in; q.c = it.g_w; q.ldc = (int)it.k_in;
    q.M = (int)it.n_out; q.N = (int)it.k_in; q.nk = (int)(it.rows / BK); q.tiles_n = (int)tn;
    grp.tile_base[n] = (int)tiles;
    grp.step_base[n] = (int)steps;
    tiles += tm * tn;
    steps += tm * tn * (it.rows / BK);
    if (tiles > 0x3fffffffLL || steps > 0x3fffffffLL) return stlt_set_error(STLT_EINVAL, "weight_grad_group: too many k-steps");
    flops += 2.0 * (double)it.n_out * (double)it.k_in * (double)it.rows;
    ++n;
  }
  if (n == 0) return 0;
  grp.n = n;
  for (int i = n; i <= STLT_GEMM_GROUP_MAX; ++i) { grp.tile_base[i] = (int)tiles; grp.step_base[i] = (int)steps; }  // sentinels: scans stop at the end
  int64_t G = n_cu() < STLT_GEMM_SK_MAX_WG ? n_cu() : STLT_GEMM_SK_MAX_WG;
  if (steps < 4 * G) G = (steps + 3) / 4;
  const int S = (int)((steps + G - 1) / G);
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(flops);
  stlt_prof_note("gemm(dW group) products=%d tiles=%lld ksteps=%lld tile=256x128 stream-K wg=%lld ksteps/wg=%d (+fix-up)", n, (long long)tiles, (long long)steps, (long long)G, S);
  float* P = t_gemm_scratch;
  hipLaunchKernelGGL((gemm_nt_kernel<STLT_ACT_NONE, false, true, true, true, true, true, true>), dim3((unsigned)G), dim3(GEMM_THREADS_WS), 0, s,
                     (const float*)nullptr, (int64_t)0, (const float*)nullptr, (int64_t)0, (const float*)nullptr, (const float*)nullptr, (int64_t)0,
                     (float*)nullptr, (int64_t)0, (int64_t)0, 0, 0, BK, 0, 0, 1, P, (unsigned long long*)nullptr, SkPlan{}, grp, SkNone{});
  if (int e = stlt_check_launch("gemm_nt_kernel(grouped stream-k)")) return e;
  hipLaunchKernelGGL(gemm_fixup_group_kernel, dim3((unsigned)(tiles * FIXUP_CHUNKS)), dim3(256), 0, s, P, S, grp);
  return stlt_check_launch("gemm_fixup_group_kernel");
}

StltGemmScratch::StltGemmScratch(void* p, size_t bytes) : prev_(t_gemm_scratch), prev_bytes_(t_gemm_scratch_bytes) {
  t_gemm_scratch = static_cast<float*>(p);
  t_gemm_scratch_bytes = p ? bytes : 0;
}
StltGemmScratch::~StltGemmScratch() {
  t_gemm_scratch = prev_;
  t_gemm_scratch_bytes = prev_bytes_;
}
StltGemmWgCap::StltGemmWgCap(int n) : prev_(t_gemm_wg_cap) { t_gemm_wg_cap = n > 0 ? (n + 7) / 8 * 8 : 0; }  // multiples of 8: the XCD-contiguous tile order
StltGemmWgCap::~StltGemmWgCap() { t_gemm_wg_cap = prev_; }
void stlt_gemm_set_scratch_impl(void* p, size_t bytes) {
  t_gemm_scratch = static_cast<float*>(p);
  t_gemm_scratch_bytes = p ? bytes : 0;
}

int launch_reduce_slabs(const float* slabs, int64_t stride, int n_slabs, float* dst, int64_t n, int accumulate, hipStream_t s) {
  if (!slabs || !dst || n_slabs < 1) return stlt_set_error(STLT_EINVAL, "reduce_slabs: bad arguments");
  if (n == 0) return 0;
  if (accumulate) { int err = 0; if (reduce_defer_take(slabs, stride, n_slabs, dst, nullptr, nullptr, 1, n, s, &err)) return err; }
  if (n_slabs > 32 && n <= 16384) {
    hipLaunchKernelGGL(reduce_tall_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, slabs, stride, n_slabs, dst, n, accumulate);
    return stlt_check_launch("reduce_tall_kernel");
  }
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slabs, stride, n_slabs, dst, n, accumulate);
  return stlt_check_launch("reduce_slabs_kernel");
}

// dst_k[i] (+)= sum_s slabs[s*stride + k*n + i] for the non-null destinations k = 0..2, i < n: one launch
int launch_reduce_slabs3(const float* slabs, int64_t stride, int n_slabs, float* dst0, float* dst1, float* dst2, int64_t n, int accumulate,
                         hipStream_t s) {
  if (!slabs || n_slabs < 1) return stlt_set_error(STLT_EINVAL, "reduce_slabs: bad arguments");
  if (n == 0 || (!dst0 && !dst1 && !dst2)) return 0;
  if (accumulate && n % 16 == 0) { int err = 0; if (reduce_defer_take(slabs, stride, n_slabs, dst0, dst1, dst2, 3, n, s, &err)) return err; }
  if (n % 16 != 0 || n > 16384) {  // shapes the fused kernel does not take: one launch per destination
    float* dsts[3] = {dst0, dst1, dst2};
    for (int k = 0; k < 3; ++k)
      if (dsts[k]) { if (int e = launch_reduce_slabs(slabs + k * n, stride, n_slabs, dsts[k], n, accumulate, s)) return e; }
    return 0;
  }
  hipLaunchKernelGGL(reduce_tall3_kernel, dim3((unsigned)(3 * n / 16)), dim3(256), 0, s, slabs, stride, n_slabs, dst0, dst1, dst2, n, accumulate);
  return stlt_check_launch("reduce_tall3_kernel");
}

int launch_linear(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int64_t M,
                  int64_t N, int64_t K, int act, hipStream_t s) {
  bool taken = false;  // opt-in split-bf16 build of the same product (gemm_bf16x3.hip); off unless STLT_GEMM_SPLIT_BF16=6
  if (int e = launch_linear_bf16x3(x, ldx, w, K, bias, nullptr, 0, y, ldy, M, N, K, act, s, &taken)) return e;
  if (taken) return 0;
  if (int e = launch_gemm_skinny(0, x, ldx, w, K, bias, nullptr, 0, y, ldy, M, N, K, act, s, &taken)) return e;  // a few rows (one per clip of a small batch): split-k partial tiles
  if (taken) return 0;
  if (int e = launch_linear_gemm16(x, ldx, w, K, bias, nullptr, 0, y, ldy, M, N, K, act, s, &taken)) return e;  // under-filled launches: whole small tiles
  if (taken) return 0;
  return launch_gemm(0, 0, x, ldx, w, K, bias, nullptr, 0, y, ldy, 0, M, N, K, 1, act, s);
}

// y = (x·Wᵀ + b) + r: the residual add of a post-norm layer in the product's epilogue (same rounding sequence as
// storing the product and adding r in the LayerNorm pass: the accumulators start from b, r is added last)
int launch_linear_add(const float* x, int64_t ldx, const float* w, const float* bias, const float* r, int64_t ldr, float* y,
                      int64_t ldy, int64_t M, int64_t N, int64_t K, hipStream_t s) {
  bool taken = false;
  if (int e = launch_linear_bf16x3(x, ldx, w, K, bias, r, ldr, y, ldy, M, N, K, STLT_ACT_NONE, s, &taken)) return e;
  if (taken) return 0;
  if (int e = launch_gemm_skinny(0, x, ldx, w, K, bias, r, ldr, y, ldy, M, N, K, STLT_ACT_NONE, s, &taken)) return e;
  if (taken) return 0;
  if (int e = launch_linear_gemm16(x, ldx, w, K, bias, r, ldr, y, ldy, M, N, K, STLT_ACT_NONE, s, &taken)) return e;
  if (taken) return 0;
  return launch_gemm(0, 0, x, ldx, w, K, bias, r, ldr, y, ldy, 0, M, N, K, 1, STLT_ACT_NONE, s);
}
