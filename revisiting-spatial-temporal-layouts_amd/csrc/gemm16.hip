// nn.Linear forward  Y = act(X·Wᵀ + b) (+ R)  for UNDER-FILLED launches: few rows (the temporal tower at the reference's default batch
// of 64 clips: M = 2048; the fusion models' 2048 / 2112-row blocks), where gemm.hip's 256 x 128 tiles are fewer than the CUs and the
// launch has to split its contraction over workgroups (stream-K: two 128-KB partial tiles per workgroup written and re-read, a
// fix-up launch, ~25 us of fixed cost on products of 40-100 us: profiles/round3_gemm_train_shapes_b64.txt, 0.32-0.67 of the MFMA peak).
//
// Here the tile is (16 RB) rows x (16 NT) columns — 128 rows x {48, 64, 96, 128, 144, 192}, round 5: 64 rows x {64, 96, 128, 160, 192, 256}
// and 32 rows x {128, 192, 256} — chosen per launch so that the number of WHOLE tiles is close to a multiple of the CU count (M = 2048:
// N = 768 -> 128 x 48 = 16 x 16 tiles; N = 2304 -> 144 wide; N = 3072 -> 192 wide; M = 5440 = 85 x 64 rows: 64 x 256 tiles = 255 / 765 /
// 1020 for N = 768 / 2304 / 3072; M = 1088 = 17 x 64: 64 x 64 = 204): every workgroup owns complete outputs, no partial tiles, no second launch.  Narrow tiles fetch more operand bytes per
// FLOP ((128 + 48) x 128 B per k-step against 0.39 MFLOP: ~15 B/clk, above the CU's ~10 B/clk), so they run fetch-bound at ~0.7 of the
// tile's MFMA rate — still well ahead of the split launch they replace.
//
// Structure (gemm16_kernel.h) = mhsa.hip's product phase: v_mfma_f32_16x16x4_f32, 8 MFMA waves each owning one 16-row block and the column tiles of its column group
// (transposed accumulators: lane = row, registers = 4 consecutive columns -> 16-byte stores), 4 DMA-only loader waves two to five
// k-steps ahead (LDS-DMA with the source-side bank swizzle, three to six stages, counted vmcnt, one barrier per k-step), bias as the accumulators'
// initial value from LDS strips DMA'd in front of each tile's first k-step, persistent workgroups over XCD-contiguous tile ranges (column tile fastest, so
// that the workgroups of an XCD share X row panels).  NT (forward) layout only; K % 32 == 0, N % 4 == 0.
#include <cstdlib>
#include "gemm16_kernel.h"

using g16::Gemm16Args;
using g16::QK;

int launch16_rb8(int nt, const Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s) {
  switch (nt) {
    case 3: return g16::launch16_nt<8, 3>(a, act, add, wkn, s);
    case 4: return g16::launch16_nt<8, 4>(a, act, add, wkn, s);
    case 6: return g16::launch16_nt<8, 6>(a, act, add, wkn, s);
    case 8: return g16::launch16_nt<8, 8>(a, act, add, wkn, s);
    case 9: return g16::launch16_nt<8, 9>(a, act, add, wkn, s);
    default: return g16::launch16_nt<8, 12>(a, act, add, wkn, s);
  }
}

namespace {

constexpr int NT_RB8[] = {3, 4, 6, 8, 9, 12};
constexpr int NT_RB4[] = {4, 6, 8, 10, 12, 16};
constexpr int NT_RB2[] = {8, 12, 16};

// tile code used between the routing and the launchers: (RB << 5) | NT, 0 = none
constexpr int tile_code(int rb, int nt) { return (rb << 5) | nt; }
constexpr int tile_rb(int code) { return code >> 5; }
constexpr int tile_nt(int code) { return code & 31; }

bool tile_ok(int rb, int nt) {
  if (rb == 8) for (int x : NT_RB8) if (x == nt) return true;
  if (rb == 4) for (int x : NT_RB4) if (x == nt) return true;
  if (rb == 2) for (int x : NT_RB2) if (x == nt) return true;
  return false;
}

int launch16_any(int code, const Gemm16Args& a, int act, bool add, bool wkn, hipStream_t s) {
  {
    const long long tiles = (long long)a.tiles_m * a.tiles_n, cus = stlt_device_cus();
    stlt_prof_note("gemm16%s M=%d N=%d K=%d act=%d%s tile=%dx%d tiles=%lld wg=%lld rounds=%lld ksteps=%d", wkn ? "(dX,W as it lies)" : "", a.M, a.N, a.K, act, add ? "+R" : "",
                   16 * tile_rb(code), 16 * tile_nt(code), tiles, tiles < cus ? tiles : cus, (tiles + cus - 1) / cus, a.K / 32);
  }
  switch (tile_rb(code)) {
    case 8: return launch16_rb8(tile_nt(code), a, act, add, wkn, s);
    case 4: return launch16_rb4(tile_nt(code), a, act, add, wkn, s);
    default: return launch16_rb2(tile_nt(code), a, act, add, wkn, s);
  }
}

// Launch-time estimate of a small-tile launch (us), per tile shape and build:  a + rounds x (k-steps x s + e)  with rounds = tiles / CUs rounded
// up: a = launch boundary + pipeline fill, s = one k-step, e = a tile's epilogue and restart.  Fitted by tools/fit_gemm16.py (weighted least
// squares) to stand-alone measurements of every tile on the shapes of profiles/round5_gemm16_shapes.txt (M = 1088 ... 16896; mean error of the fit 1 - 2 %
// forward, 3 - 6 % input gradient; the k-steps run at 0.80 - 0.94 of the matrix pipe's rate forward, 0.73 - 0.88 with the [k][n] gather).
struct TileCost { int rb, nt; float a, s, e; };
constexpr TileCost COST_FWD[] = {
  {8, 3, 3.49f, 0.7074f, 1.76f},
  {8, 4, 4.60f, 0.9567f, 1.57f},
  {8, 6, 4.59f, 1.3932f, 2.06f},
  {8, 8, 4.66f, 1.8414f, 2.27f},
  {8, 9, 4.96f, 2.0490f, 2.68f},
  {8, 12, 4.36f, 2.7202f, 3.48f},
  {4, 4, 4.25f, 0.5411f, 0.43f},
  {4, 6, 4.79f, 0.7264f, 0.89f},
  {4, 8, 5.39f, 0.9631f, 1.25f},
  {4, 10, 5.20f, 1.1681f, 1.57f},
  {4, 12, 4.38f, 1.4250f, 1.50f},
  {4, 16, 4.03f, 1.8458f, 2.54f},
  {2, 8, 4.41f, 0.5371f, 0.34f},
  {2, 12, 4.48f, 0.7329f, 0.89f},
  {2, 16, 4.72f, 0.9653f, 1.33f},
};
constexpr TileCost COST_WKN[] = {
  {8, 3, 4.30f, 0.7784f, 2.48f},
  {8, 4, 4.86f, 1.0441f, 2.75f},
  {8, 6, 4.76f, 1.5201f, 3.86f},
  {8, 8, 5.09f, 1.9871f, 4.80f},
  {8, 9, 5.92f, 2.1656f, 4.35f},
  {8, 12, 5.38f, 2.9609f, 7.64f},
  {4, 4, 6.33f, 0.5995f, 1.31f},
  {4, 6, 6.79f, 0.7909f, 1.84f},
  {4, 8, 5.82f, 1.0507f, 2.71f},
  {4, 10, 5.22f, 1.2805f, 3.73f},
  {4, 12, 4.90f, 1.5506f, 3.84f},
  {4, 16, 5.13f, 2.0014f, 5.04f},
  {2, 8, 6.36f, 0.5920f, 1.35f},
  {2, 12, 6.24f, 0.7955f, 1.90f},
  {2, 16, 5.12f, 1.0516f, 2.95f},
};
double est16_us(int64_t M, int64_t N, int64_t K, int rb, int nt, int64_t cus, bool wkn = false) {
  const int64_t tiles = ((M + 16 * rb - 1) / (16 * rb)) * ((N + 16 * nt - 1) / (16 * nt));
  const int64_t rounds = (tiles + cus - 1) / cus;
  for (const TileCost& c : (wkn ? COST_WKN : COST_FWD))
    if (c.rb == rb && c.nt == nt) return (double)c.a + (double)rounds * ((double)(K / QK) * (double)c.s + (double)c.e);
  return 1e30;  // not a tile of the kernel
}
// gemm.hip's launch: 256 x 128 tiles; whole-tile rounds when they fill >= 0.9 of the last round (3.70 us per k-step + 4 us per round:
// 94 / 360 us per round at K = 768 / 3072, profiles/round4_gemm_shapes_b1024.txt), else equal k-step shares (stream-K) + the fixed cost of
// the partial tiles and the fix-up launch (24 us + 0.1 us per tile below one round, ~30 us for the tail of a longer launch)
double est_big_us(int64_t M, int64_t N, int64_t K, int64_t cus, bool nn = false, bool* whole = nullptr) {  // nn: the input-gradient (NN) build, ~6 % slower per k-step
  const int64_t tiles = ((M + 255) / 256) * ((N + 127) / 128);
  const int64_t rounds = (tiles + cus - 1) / cus;
  const double nk = (double)(K / QK);
  const double fill = (double)tiles / (double)(rounds * cus);
  if (whole) *whole = fill >= 0.9;
  if (fill >= 0.9) return (double)rounds * (nk * (nn ? 3.92 : 3.70) + 4.0);
  return (double)tiles * nk / (double)cus * (nn ? 3.85 : 3.62) + (tiles < cus ? 24.0 + 0.1 * (double)tiles : 30.0);
}

bool shape_ok(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw) {
  // (the loaders address a tile's rows with 32-bit byte offsets from the tile's origin: pitches below 4M floats)
  return M > 0 && N > 0 && K >= 2 * QK && K % QK == 0 && N % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0 && M <= 0x3fffff00LL && N <= 0x3fffff00LL &&
         ldx <= 0x3fffffLL && ldw <= 0x3fffffLL;
}

}  // namespace

// Public tile parameter of the C-ABI (stlt_linear_small_fwd's tile_cols and the *_choice results): columns | rows << 16, rows 0 = 128.
int stlt_gemm16_tile_from_public(int tile) {
  const int cols = tile & 0xffff, rows = (tile >> 16) ? (tile >> 16) : 128;
  if (cols % 16 != 0 || rows % 16 != 0 || !tile_ok(rows / 16, cols / 16)) return 0;
  return tile_code(rows / 16, cols / 16);
}
int stlt_gemm16_tile_to_public(int code) {
  if (code == 0) return 0;
  const int rows = 16 * tile_rb(code), cols = 16 * tile_nt(code);
  return rows == 128 ? cols : (cols | (rows << 16));
}

// Would launch_linear hand this product to the small-tile kernel, and with which tile?  0: no (gemm.hip keeps it), else the tile code.
// STLT_GEMM16=0 switches the kernel off, STLT_GEMM16=1 forces it onto every product it can take (A/B runs); otherwise the two
// launch-time estimates above decide.  STLT_GEMM16_ROWS=128|64|32 restricts the tile heights considered (A/B runs).
static int g_gemm16_mode = -2;  // -2: not read yet; -1: by estimate; 0: off; 1: every product the kernel can take
static int g_gemm16_rows = -1;  // tile heights considered: 0 = all, 128 / 64 / 32 = that one only; -1: not read yet (STLT_GEMM16_ROWS)
int stlt_gemm16_set_mode(int mode) {
  if (mode == 128 || mode == 64 || mode == 32) { g_gemm16_mode = -1; g_gemm16_rows = mode; return 0; }  // by estimate, one tile height (tests, A/B runs)
  if (mode < -2 || mode > 1)
    return stlt_set_error(STLT_EINVAL, "small-tile products: mode -1 (by estimate), 0 (off), 1 (always), 128 / 64 / 32 (by estimate, tiles of that height only) or -2 (back to STLT_GEMM16 / the default)");
  g_gemm16_mode = mode;  // -2: the next routing decision re-reads the environment
  g_gemm16_rows = mode == -2 ? -1 : 0;
  return 0;
}
int stlt_gemm16_choice(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, bool wkn) {
  if (g_gemm16_mode == -2) { const char* e = getenv("STLT_GEMM16"); g_gemm16_mode = e ? (atoi(e) == 0 ? 0 : (atoi(e) == 1 ? 1 : -1)) : -1; }
  const int mode = g_gemm16_mode;
  static const int force_nt = [] { const char* e = getenv("STLT_GEMM16_NT"); return e ? atoi(e) : 0; }();
  if (g_gemm16_rows < 0) { const char* e = getenv("STLT_GEMM16_ROWS"); g_gemm16_rows = e ? atoi(e) : 0; }
  const int only_rows = g_gemm16_rows;
  if (mode == 0) return 0;
  if (!shape_ok(M, N, K, ldx, ldw)) return 0;
  const int64_t cus = stlt_device_cus();
  int best = 0;
  double best_us = 1e30;
  auto consider = [&](int rb, int nt) {
    if (force_nt && nt != force_nt) return;
    if (only_rows && 16 * rb != only_rows) return;
    const int64_t tiles = ((M + 16 * rb - 1) / (16 * rb)) * ((N + 16 * nt - 1) / (16 * nt));
    if (tiles > 0x3fffffffLL) return;
    const double us = est16_us(M, N, K, rb, nt, cus, wkn);
    if (us < best_us) { best_us = us; best = tile_code(rb, nt); }
  };
  for (int nt : NT_RB8) consider(8, nt);  // ties keep the taller tile (less operand fetch per FLOP)
  for (int nt : NT_RB4) consider(4, nt);
  for (int nt : NT_RB2) consider(2, nt);
  if (best == 0) return 0;
  if (mode == 1) return best;
  // Against a stream-K launch a tie goes to the small tiles (no partial tiles in the caches, no fix-up launch beside the training sweep's
  // side-stream products); against whole-tile rounds of the large kernel they must win by 3 % (the two kernels are within 1 - 2 % of each
  // other per FLOP there — profiles/round5_gemm16_shapes.txt, M = 32768 / 229376 — and the estimates are not better than that)
  bool whole = false;
  const double big = est_big_us(M, N, K, cus, wkn, &whole);
  return best_us < (whole ? 0.97 : (wkn ? 1.03 : 1.0)) * big ? best : 0;
}

// Estimated duration (us) of the nn.Linear forward launch_linear would make for this shape: the faster of the two kernels' estimates
// (what the routing picks).  Used by the fused MHSA dispatch to price the product + attention-core pair it competes with.
double stlt_linear_est_us(int64_t M, int64_t N, int64_t K) {
  const int64_t cus = stlt_device_cus();
  double us = est_big_us(M, N, K, cus);
  const int code = stlt_gemm16_choice(M, N, K, K, K, false);
  if (code > 0) { const double small = est16_us(M, N, K, tile_rb(code), tile_nt(code), cus); if (small < us) us = small; }
  return us;
}

// Y (M, N; ldy) = act(X (M, K; ldx) · W (N, K; ldw)ᵀ + bias) (+ R (ldr)) on the small-tile kernel; *taken = false when the shape is not its.
// force_tile: a tile code (tests, A/B runs; 0 = by the routing).
int launch_linear_gemm16(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* r, int64_t ldr, float* y, int64_t ldy,
                         int64_t M, int64_t N, int64_t K, int act, hipStream_t s, bool* taken, int force_tile) {
  *taken = false;
  if (act != STLT_ACT_NONE && act != STLT_ACT_GELU && act != STLT_ACT_RELU) return 0;
  if (r && act != STLT_ACT_NONE) return 0;
  int code = force_tile;
  if (code == 0) code = stlt_gemm16_choice(M, N, K, ldx, ldw, false);
  else if (!shape_ok(M, N, K, ldx, ldw) || !tile_ok(tile_rb(code), tile_nt(code)))
    return stlt_set_error(STLT_EINVAL, "gemm16: K must be a multiple of 32 (>= 64), N and the row pitches multiples of 4, tiles of 128 x 16{3,4,6,8,9,12}, 64 x 16{4,6,8,10,12,16} or 32 x 16{8,12,16}");
  if (code == 0) return 0;
  if (!x || !w || !y) return stlt_set_error(STLT_EINVAL, "gemm16: null pointer");
  if (ldx < K || ldw < K || ldy < N || (r && ldr < N) || ldy % 4 != 0 || (r && ldr % 4 != 0))
    return stlt_set_error(STLT_EINVAL, "gemm16: bad leading dimension (ldx=%lld ldw=%lld ldy=%lld ldr=%lld)", (long long)ldx, (long long)ldw, (long long)ldy, (long long)ldr);
  const int rb = tile_rb(code), nt = tile_nt(code);
  Gemm16Args a{};
  a.X = x; a.W = w; a.bias = bias; a.R = r; a.Y = y;
  a.ldx = ldx; a.ldw = ldw; a.ldr = ldr; a.ldy = ldy;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)((M + 16 * rb - 1) / (16 * rb));
  a.tiles_n = (int)((N + 16 * nt - 1) / (16 * nt));
  if ((int64_t)a.tiles_m * a.tiles_n > 0x3fffffffLL) return stlt_set_error(STLT_EINVAL, "gemm16: too many tiles");
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)M * (double)N * (double)K);
  *taken = true;
  return launch16_any(code, a, act, r != nullptr, false, s);
}

int launch_linear_gelu_keep(const float* x, int64_t ldx, const float* w, const float* bias, float* u, float* h, int64_t M, int64_t N, int64_t K,
                            StltDrop dr, uint32_t site, const int* drop_rows, hipStream_t s) {
  if (!x || !w || !u || !h) return stlt_set_error(STLT_EINVAL, "linear + GELU (kept pre-activation): null pointer");
  static const bool fused_on = [] { const char* e = getenv("STLT_FFN1_KEEP_FUSED"); return !(e && e[0] == '0'); }();
  const int code = fused_on && !stlt_split_bf16_takes(M, N, K, ldx, K) ? stlt_gemm16_choice(M, N, K, ldx, K, false) : 0;
  if (code == 0) {  // large tiles (or the small-tile kernel switched off): the product, then the element-wise pass over u
    if (int e = launch_linear(x, ldx, w, bias, u, N, M, N, K, STLT_ACT_NONE, s)) return e;
    return launch_gelu_fwd(u, h, M * N, s, dr, site, drop_rows, N);
  }
  const int rb = tile_rb(code), nt = tile_nt(code);
  Gemm16Args a{};
  a.epi.dr = dr; a.epi.site = site; a.epi.drop_rows = drop_rows; a.epi.cs_part = nullptr;
  a.X = x; a.W = w; a.bias = bias; a.R = u; a.Y = h;
  a.ldx = ldx; a.ldw = K; a.ldr = N; a.ldy = N;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)((M + 16 * rb - 1) / (16 * rb));
  a.tiles_n = (int)((N + 16 * nt - 1) / (16 * nt));
  if ((int64_t)a.tiles_m * a.tiles_n > 0x3fffffffLL) return stlt_set_error(STLT_EINVAL, "gemm16: too many tiles");
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)M * (double)N * (double)K);
  return launch16_any(code, a, STLT_ACT_GELU_KEEP, false, false, s);
}

// C (rows, k_in; ldc) = dY (rows, n_out; ld_dy) · W (n_out, k_in) (+ R): the input gradient of a Linear on the small-tile kernel, W read
// as it lies (WKN build).  *taken = false: the product stays on gemm.hip's NN kernel (shape not taken, or estimated slower).
int launch_input_grad_gemm16(const float* dy, int64_t ld_dy, const float* w, int64_t n_out, int64_t k_in, const float* r, int64_t ldr, float* c,
                             int64_t ldc, int64_t rows, hipStream_t s, bool* taken, int force_tile, const StltGemmEpi* gelu_bwd) {
  // gelu_bwd != null: the FFN hidden gradient in the epilogue — c = drop(dy·w) ∘ gelu'(r) with r the pre-activation (not added), and the
  // column sums of c left in gelu_bwd->cs_part (16 partial rows per 256 rows; train.hip: stlt_ffn_hidden_backward_fused)
  *taken = false;
  if (gelu_bwd && (!r || !gelu_bwd->cs_part)) return stlt_set_error(STLT_EINVAL, "gemm16 (GELU backward): the pre-activation and a column-sum buffer are required");
  // a current transposed copy of the weight (a trainer step refreshed it: wt_cache.hip): the product is a forward product dX = dY·(Wt)ᵀ on
  // the forward build — same add-source / GELU-backward epilogues, the forward build's cost table for the tile
  const float* wt = nullptr;
  int64_t ldwt = 0;
  if (force_tile == 0 && dy && w && c && stlt_wt_lookup(w, n_out, k_in, &wt, &ldwt)) {
    const int fcode = stlt_gemm16_choice(rows, k_in, n_out, ld_dy, ldwt, false);
    if (fcode != 0) {
      if (ld_dy < n_out || ldc < k_in || (r && ldr < k_in) || ldc % 4 != 0 || (r && ldr % 4 != 0)) return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): bad leading dimension");
      const int frb = tile_rb(fcode), fnt = tile_nt(fcode);
      Gemm16Args a{};
      if (gelu_bwd) a.epi = *gelu_bwd;
      a.X = dy; a.W = wt; a.bias = nullptr; a.R = r; a.Y = c;
      a.ldx = ld_dy; a.ldw = ldwt; a.ldr = ldr; a.ldy = ldc;
      a.M = (int)rows; a.N = (int)k_in; a.K = (int)n_out;
      a.tiles_m = (int)((rows + 16 * frb - 1) / (16 * frb));
      a.tiles_n = (int)((k_in + 16 * fnt - 1) / (16 * fnt));
      if ((int64_t)a.tiles_m * a.tiles_n > 0x3fffffffLL) return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): too many tiles");
      StltProfScope ps(STLT_K_GEMM, s);
      stlt_prof_add_flops(2.0 * (double)rows * (double)k_in * (double)n_out);
      *taken = true;
      const int rc = launch16_any(fcode, a, gelu_bwd ? STLT_ACT_GELU_BWD : STLT_ACT_NONE, r != nullptr && !gelu_bwd, false, s);
      if (rc == 0) stlt_wt_count_hit();  // served: the product was launched on the copy
      return rc;
    }
  }
  int code = force_tile;
  if (code == 0) code = stlt_gemm16_choice(rows, k_in, n_out, ld_dy, k_in, true);
  else if (!shape_ok(rows, k_in, n_out, ld_dy, k_in) || !tile_ok(tile_rb(code), tile_nt(code)))
    return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): n_out must be a multiple of 32 (>= 64), k_in and the row pitches multiples of 4");
  if (code == 0) return 0;
  if (!dy || !w || !c) return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): null pointer");
  if (ld_dy < n_out || ldc < k_in || (r && ldr < k_in) || ldc % 4 != 0 || (r && ldr % 4 != 0)) return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): bad leading dimension");
  const int rb = tile_rb(code), nt = tile_nt(code);
  Gemm16Args a{};
  if (gelu_bwd) a.epi = *gelu_bwd;
  a.X = dy; a.W = w; a.bias = nullptr; a.R = r; a.Y = c;
  a.ldx = ld_dy; a.ldw = k_in; a.ldr = ldr; a.ldy = ldc;
  a.M = (int)rows; a.N = (int)k_in; a.K = (int)n_out;
  a.tiles_m = (int)((rows + 16 * rb - 1) / (16 * rb));
  a.tiles_n = (int)((k_in + 16 * nt - 1) / (16 * nt));
  if ((int64_t)a.tiles_m * a.tiles_n > 0x3fffffffLL) return stlt_set_error(STLT_EINVAL, "gemm16 (input gradient): too many tiles");
  StltProfScope ps(STLT_K_GEMM, s);
  stlt_prof_add_flops(2.0 * (double)rows * (double)k_in * (double)n_out);
  *taken = true;
  return launch16_any(code, a, gelu_bwd ? STLT_ACT_GELU_BWD : STLT_ACT_NONE, r != nullptr && !gelu_bwd, true, s);
}
