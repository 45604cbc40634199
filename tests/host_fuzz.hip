// Host-side fuzz of libstlt_hip's launchers under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_sanitizers.py builds the
// library's sources with `hipcc --offload-host-only -fsanitize=address,undefined` and links this driver against them; no GPU is involved).
//
// What runs on the host inside the C-ABI is plan arithmetic: byte-size functions, workspace / tape / scratch layouts, the small-tile cost
// tables and routing estimates, stream-K and grouped-launch plans, persistent-grid sizing, argument checks.  This driver feeds every entry
// point random, boundary (multiples of the tile sizes +- 1) and extreme shapes (row counts up to 2^31, hidden sizes up to 4096, negative
// and zero dimensions, null and misaligned pointers) with FAKE device pointers: the host side never dereferences a device pointer (ASan
// turns one that does into a crash), signed overflow / bad shifts / out-of-range indices in the plan arithmetic trip UBSan, and overruns
// of the host-side plan tables trip ASan.  The HIP runtime is tests/host_fuzz_stubs.cpp: it checks every launch configuration as the real one
// does (empty or oversized grids, > 1024 threads, > 160 KB LDS are refused) and then reports success, so a whole forward / reverse sweep walks
// its complete launch sequence on the host.  Exit code 0 = no sanitizer report (reports abort: -fno-sanitize-recover).
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"

extern "C" long stlt_fake_hip_launches();
extern "C" long stlt_fake_hip_refused();

namespace {

uint64_t g_state = 1;
uint64_t rnd() {
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
int64_t uni(int64_t lo, int64_t hi) { return lo + (int64_t)(rnd() % (uint64_t)(hi - lo + 1)); }
bool coin(int one_in) { return rnd() % (uint64_t)one_in == 0; }

// a dimension: mostly the sizes the path sees, sometimes a tile boundary +- 1, sometimes extreme / degenerate
int64_t dim(int64_t typical_max, int64_t extreme_max) {
  switch (rnd() % 10) {
    case 0: return 0;
    case 1: return coin(2) ? -uni(1, 1000) : 1;
    case 2: { static const int64_t b[] = {16, 32, 64, 96, 128, 144, 192, 256, 768, 1024, 3072};
              return b[rnd() % 11] * uni(1, 8) + uni(-1, 1); }
    case 3: return uni(1, extreme_max);
    case 4: return extreme_max - uni(0, 3);
    default: return uni(1, typical_max);
  }
}
int64_t hidden() {  // hidden sizes: multiples of 64 mostly, anything up to 4096 sometimes
  switch (rnd() % 8) {
    case 0: return uni(1, 4096);
    case 1: return 4 * uni(1, 1024);
    case 2: return 0;
    default: return 64 * uni(1, 16);
  }
}

// Fake device memory: addresses in an unmapped range, 256-byte aligned, distinct per call site.  Never touched by a correct host path.
char* fake(int slot, bool misalign = false) { return (char*)(uintptr_t)(0x7e0000000000ull + (uint64_t)slot * 0x40000000ull + (misalign ? 4 : 0)); }
template <typename T> T* F(int slot) { return coin(40) ? nullptr : (T*)fake(slot, coin(25)); }
template <typename T> T* FA(int slot) { return (T*)fake(slot); }  // always valid-looking and aligned

long g_calls = 0, g_ok = 0;
void tally(int rc) { ++g_calls; g_ok += rc == 0; }

void fuzz_bytes() {
  const int64_t B = dim(4096, 1ll << 22), T = dim(64, 512), N = dim(36, 128), d = hidden(), K = dim(174, 100000);
  (void)stlt_workspace_bytes(B, T, N, d, K);
  (void)stlt_caf_workspace_bytes(B, T, N, d, dim(2048, 1 << 16), dim(32, 4096), K);
  (void)stlt_train_tape_bytes(B, T, N, d, dim(4, 64), dim(8, 64));
  (void)stlt_train_scratch_bytes(B, T, N, d, dim(38, 4096));
  (void)stlt_linear_bwd_scratch_bytes(dim(3072, 1ll << 31));
  (void)stlt_attn_core_bwd_scratch_bytes(dim(12, 1 << 20));
  (void)stlt_add_layernorm_bwd_scratch_bytes(d);
  (void)stlt_embed_bwd_scratch_bytes(dim(1 << 20, 1ll << 31), dim(38, 4096), d);
  (void)stlt_frames_embed_bwd_scratch_bytes(T, d);
  (void)stlt_block_keep_bytes(dim(1 << 16, 1ll << 31), d, (int)uni(-1, 2));
  (void)stlt_block_work_bytes(dim(1 << 16, 1ll << 31), d);
  (void)stlt_gemm_scratch_bytes();
  (void)stlt_debug_buffer_bytes();
  (void)stlt_eval_max_clips();
  g_calls += 14;
}

void fuzz_estimates() {
  const int64_t M = dim(1 << 16, 1ll << 31), N = dim(3072, 1 << 20), K = dim(3072, 1 << 20);
  (void)stlt_linear_small_choice(M, N, K);
  (void)stlt_input_grad_small_choice(M, N, K);
  (void)stlt_linear_est_us(M, N, K);
  (void)stlt_gemm16_choice(M, N, K, K + 4 * uni(0, 4), K + 4 * uni(0, 4), coin(2));
  (void)stlt_gemm16_tile_from_public((int)rnd());
  (void)stlt_gemm16_tile_to_public((int)uni(0, 4096));
  (void)stlt_fused_mhsa_active(dim(64, 512), hidden(), dim(12, 64));
  (void)stlt_fused_mhsa_used(dim(1 << 16, 1ll << 31), dim(64, 512), hidden(), dim(12, 64), (int)uni(0, 1));
  (void)stlt_split_bf16_takes(M, N, K, K, K);
  g_calls += 9;
}

void fuzz_products() {
  const int64_t M = dim(1 << 16, 1ll << 31), N = dim(3072, 1 << 16), K = coin(4) ? dim(3072, 1 << 16) : 32 * uni(1, 128);
  const int act = (int)uni(-1, 4);
  void* s = nullptr;
  tally(stlt_linear_fwd(F<float>(1), K + 4 * uni(0, 2), F<float>(2), F<float>(3), F<float>(4), N + 4 * uni(0, 2), M, N, K, act, s));
  const int tile = coin(3) ? (int)rnd() : (int)((uni(0, 3) * 32) << 16 | 16 * uni(1, 16));
  tally(stlt_linear_small_fwd(F<float>(1), K, F<float>(2), F<float>(3), coin(2) ? nullptr : FA<float>(5), N, F<float>(4), N, M, N, K, act, tile, s));
  tally(stlt_input_grad_small(F<float>(1), N, F<float>(2), N, K, coin(2) ? nullptr : FA<float>(5), K, F<float>(4), K, M, coin(3) ? 0 : tile, nullptr, s));
  tally(stlt_gemm((int)uni(0, 1), (int)uni(0, 1), F<float>(1), dim(3072, 1 << 20), F<float>(2), dim(3072, 1 << 20), coin(2) ? nullptr : FA<float>(3),
                  dim(3072, 1 << 20), F<float>(4), dim(3072, 1 << 20), dim(1 << 20, 1ll << 40), M, N, K, (int)uni(-1, 40), s));
  // with lent stream-K scratch the under-filled launches build their equal-k-step plans
  (void)stlt_gemm_set_scratch(FA<char>(30), stlt_gemm_scratch_bytes());
  tally(stlt_linear_fwd(FA<float>(1), K, FA<float>(2), FA<float>(3), FA<float>(4), N, M, N, K, (int)uni(0, 2), s));
  tally(stlt_gemm(1, 1, FA<float>(1), N, FA<float>(2), K, FA<float>(4), K, FA<float>(4), K, 0, N, K, 32 * uni(1, 512), 1, s));  // a weight gradient: dw += dyT x
  tally(stlt_gemm(0, 1, FA<float>(1), N, FA<float>(2), K, nullptr, 0, FA<float>(4), K, 0, M, K, 32 * uni(1, 96), 1, s));          // an input gradient
  stlt_wgrad_item items[40];
  const int n_items = (int)uni(-1, 36);
  for (int i = 0; i < 40; ++i) items[i] = stlt_wgrad_item{FA<float>(6), 64 * uni(1, 48), FA<float>(7), 64 * uni(1, 48), coin(5) ? dim(4096, 1 << 20) : 32 * uni(1, 128), FA<float>(8 + i % 4)};
  tally(stlt_weight_grad_group(coin(30) ? nullptr : items, n_items, s));
  (void)stlt_gemm_set_scratch(nullptr, 0);
  tally(stlt_reduce_slabs(F<float>(1), dim(1 << 20, 1ll << 40), (int)uni(-1, 64), F<float>(2), dim(1 << 20, 1ll << 33), (int)uni(0, 1), s));
  (void)stlt_set_gemm_small_tiles((int)uni(-3, 2));
  (void)stlt_set_gemm_split_bf16(coin(4) ? (int)uni(-1, 8) : 0);
}

void fuzz_rowwise_and_attention() {
  void* s = nullptr;
  const int64_t S = dim(1 << 15, 1ll << 31), L = dim(64, 2048), H = dim(12, 64), dh = coin(3) ? dim(64, 300) : 64, d = hidden();
  tally(stlt_attn_core_fwd(F<float>(1), F<uint8_t>(2), (int)uni(0, 1), S, L, H, dh, F<float>(3), s));
  tally(stlt_mhsa_fused_fwd(F<float>(1), F<float>(2), F<float>(3), F<uint8_t>(4), S, L, H, d, F<float>(5), s));
  tally(stlt_mhsa_fused_fwd_ex(F<float>(1), F<float>(2), F<float>(3), F<uint8_t>(4), (int)uni(0, 1), S, L, H, coin(2) ? 64 * H : d, coin(2) ? 0.f : 0.1f, rnd(),
                               (uint32_t)rnd(), F<float>(5), coin(2) ? nullptr : FA<float>(6), s));
  const int64_t Lk = dim(64, 2048);
  tally(stlt_attn_cross_fwd(F<float>(1), dim(768, 1 << 16), F<float>(2), F<float>(3), dim(1536, 1 << 16), F<uint8_t>(4), (int)uni(0, 1), S, L, Lk, H, dh, F<float>(5), s));
  tally(stlt_attn_ragged_fwd(F<float>(1), F<int32_t>(2), F<int32_t>(3), (int)uni(0, 1), dim(1 << 18, 1ll << 31), H, dh, F<float>(4), s));
  tally(stlt_attn_fwd_dropout(F<float>(1), dim(768, 1 << 16), F<float>(2), F<float>(3), dim(1536, 1 << 16), F<uint8_t>(4), (int)uni(0, 1), S, L, Lk, H, dh,
                              coin(8) ? 1.5f : 0.1f, rnd(), (uint32_t)rnd(), F<float>(5), s));
  tally(stlt_attn_bwd(F<float>(1), dim(768, 1 << 16), F<float>(2), F<float>(3), dim(1536, 1 << 16), F<float>(4), F<uint8_t>(5), (int)uni(0, 1), S, L, Lk, H, dh,
                      coin(8) ? -0.5f : 0.1f, rnd(), (uint32_t)rnd(), F<float>(6), dim(768, 1 << 16), F<float>(7), F<float>(8), dim(1536, 1 << 16), s));
  tally(stlt_attn_core_bwd(F<float>(1), F<float>(2), F<uint8_t>(3), (int)uni(0, 1), S, L, H, dh, 0.1f, rnd(), (uint32_t)rnd(), F<float>(4), coin(2) ? nullptr : FA<float>(5),
                           F<char>(6), coin(3) ? (size_t)uni(0, 1 << 20) : stlt_attn_core_bwd_scratch_bytes(H), s));
  const int64_t M = dim(1 << 18, 1ll << 31);
  tally(stlt_add_layernorm_fwd(F<float>(1), d, coin(2) ? nullptr : FA<float>(2), d, F<float>(3), F<float>(4), 1e-5f, M, d, F<float>(5), d, s));
  tally(stlt_add_layernorm_bwd(F<float>(1), F<float>(2), coin(2) ? nullptr : FA<float>(3), F<float>(4), 1e-5f, M, d, F<float>(5), F<float>(6), F<float>(7), F<char>(8),
                               coin(3) ? (size_t)uni(0, 1 << 20) : stlt_add_layernorm_bwd_scratch_bytes(d), s));
  const int64_t tok = dim(1 << 20, 1ll << 31), ncat = dim(38, 4096);
  tally(stlt_embed_fwd(F<int64_t>(1), F<float>(2), coin(2) ? nullptr : FA<float>(3), F<float>(4), ncat, F<float>(5), F<float>(6), F<float>(7), F<float>(8), F<float>(9),
                       F<float>(10), 1e-12f, tok, d, F<float>(11), s));
  tally(stlt_embed_fwd_train(F<int64_t>(1), F<float>(2), coin(2) ? nullptr : FA<float>(3), F<float>(4), ncat, F<float>(5), F<float>(6), F<float>(7), F<float>(8), F<float>(9),
                             F<float>(10), 1e-12f, tok, d, F<float>(11), F<float>(12), s));
  tally(stlt_embed_bwd(F<float>(1), F<int64_t>(2), F<float>(3), coin(2) ? nullptr : FA<float>(4), ncat, tok, d, F<float>(5), F<float>(6), F<float>(7), F<float>(8), F<float>(9),
                       F<char>(10), coin(3) ? (size_t)uni(0, 1 << 20) : stlt_embed_bwd_scratch_bytes(tok, ncat, d), s));
  const int64_t B = dim(1024, 1ll << 24), T = dim(64, 512);
  tally(stlt_frames_embed_fwd(F<float>(1), dim(768 * 7, 1 << 20), F<int64_t>(2), F<float>(3), F<float>(4), F<float>(5), F<float>(6), 1e-12f, B, T, d, F<float>(7), s));
  tally(stlt_frames_embed_fwd_train(F<float>(1), dim(768 * 7, 1 << 20), F<int64_t>(2), F<float>(3), F<float>(4), F<float>(5), F<float>(6), 1e-12f, B, T, d, F<float>(7),
                                    F<float>(8), s));
  tally(stlt_frames_embed_bwd(F<float>(1), F<int64_t>(2), B, T, d, F<float>(3), F<float>(4), F<char>(5), coin(3) ? (size_t)uni(0, 1 << 20) : stlt_frames_embed_bwd_scratch_bytes(T, d), s));
  tally(stlt_gather_last_fwd(F<float>(1), F<int64_t>(2), B, T, d, F<float>(3), s));
  tally(stlt_collate_fwd(F<int64_t>(1), F<float>(2), F<float>(3), F<int64_t>(4), F<int64_t>(5), B, T, dim(36, 128), uni(0, 40), F<int64_t>(6), F<float>(7), F<float>(8),
                         F<int64_t>(9), F<uint8_t>(10), F<uint8_t>(11), s));
  const int64_t n = dim(1 << 24, 1ll << 40);
  tally(stlt_gelu_fwd(F<float>(1), F<float>(2), n, s));
  tally(stlt_gelu_bwd(F<float>(1), F<float>(2), F<float>(3), n, s));
  tally(stlt_relu_bwd(F<float>(1), F<float>(2), F<float>(3), n, s));
  tally(stlt_dropout(F<float>(1), F<float>(2), n, coin(6) ? 1.0f : 0.1f, rnd(), (uint32_t)rnd(), s));
  tally(stlt_loss_fwd_bwd(F<float>(1), F<char>(2), (int)uni(-1, 2), B, dim(174, 1 << 20), 0.25f, F<float>(3), F<float>(4), F<float>(5), s));
  tally(stlt_grad_norm(F<float>(1), n, coin(2) ? 0.f : 5.f, F<float>(2), F<float>(3), s));
  tally(stlt_adamw_step(F<stlt_opt_chunk>(1), dim(8192, 1ll << 31), F<float>(2), F<float>(3), F<float>(4), coin(2) ? nullptr : FA<float>(5), 5e-5f, 0.9f, 0.999f, 1e-8f,
                        dim(1000, 1ll << 40), s));
  tally(stlt_eval_topk(F<float>(1), dim(174, 1 << 20), F<int64_t>(2), B, dim(174, 1 << 20), F<int64_t>(3), s));
  tally(stlt_eval_store_sigmoid(F<float>(1), dim(157, 1 << 20), F<float>(2), B, dim(157, 1 << 20), F<double>(3), F<double>(4), dim(1000, 1ll << 40), s));
  tally(stlt_eval_average_precision(F<float>(1), F<float>(2), dim(2000, 1ll << 31), dim(157, 1 << 20), F<double>(3), F<double>(4), F<uint8_t>(5), s));
}

struct Model {
  std::vector<stlt_layer_params> sp, tp;
  stlt_params p{};
  stlt_inputs in{};
};
stlt_layer_params fake_layer(int base) {
  stlt_layer_params l;
  const float** f = reinterpret_cast<const float**>(&l);
  for (size_t i = 0; i < sizeof(l) / sizeof(float*); ++i) f[i] = FA<float>(base + (int)i);
  return l;
}
void make_model(Model& m, bool with_head) {
  m.p.d = coin(6) ? hidden() : 768;
  m.p.H = coin(6) ? dim(12, 64) : (m.p.d > 0 ? m.p.d / 64 : 0);
  m.p.n_categories = dim(38, 300);
  m.p.n_spatial = coin(8) ? uni(-1, 40) : uni(1, 4);
  m.p.n_temporal = coin(8) ? uni(-1, 40) : uni(1, 8);
  m.p.n_classes = dim(174, 5000);
  m.p.n_positions = coin(4) ? dim(256, 1024) : 256;
  m.p.ln_eps = 1e-12f;
  const float** f = &m.p.cat_emb;
  for (int i = 0; i < 11; ++i) f[i] = FA<float>(40 + i);
  if (coin(3)) m.p.score_w = m.p.score_b = nullptr;
  m.sp.assign((size_t)(m.p.n_spatial > 0 ? m.p.n_spatial : 0), fake_layer(60));
  m.tp.assign((size_t)(m.p.n_temporal > 0 ? m.p.n_temporal : 0), fake_layer(80));
  m.p.spatial = m.sp.empty() ? (coin(2) ? nullptr : reinterpret_cast<const stlt_layer_params*>(FA<char>(0))) : m.sp.data();
  m.p.temporal = m.tp.empty() ? nullptr : m.tp.data();
  if (m.sp.empty() && m.p.n_spatial > 0) m.p.n_spatial = 0;
  if (m.p.spatial && m.sp.empty()) m.p.spatial = nullptr;
  const float** h = &m.p.fc1_w;
  for (int i = 0; i < 6; ++i) h[i] = with_head ? FA<float>(100 + i) : nullptr;
  m.in.B = dim(1024, 1ll << 22);
  m.in.T = dim(64, 300);
  m.in.N = dim(36, 128);
  m.in.categories = F<int64_t>(110); m.in.boxes = F<float>(111); m.in.scores = coin(2) ? nullptr : FA<float>(112);
  if (coin(3)) { m.in.n_real_tokens = dim(1 << 16, 1ll << 33); m.in.n_real_frames = dim(1 << 12, 1ll << 31); }  // the caller's row counts for skip-padding
  m.in.kpm_boxes = F<uint8_t>(113); m.in.frame_types = F<int64_t>(114); m.in.kpm_frames = F<uint8_t>(115); m.in.lengths = F<int64_t>(116);
}

void fuzz_whole_path() {
  Model m;
  make_model(m, !coin(5));
  void* s = nullptr;
  const size_t need = stlt_workspace_bytes(m.in.B, m.in.T, m.in.N, m.p.d, m.p.n_classes);
  const size_t ws = coin(4) ? (size_t)uni(0, 1 << 24) : need;
  const int flags = coin(3) ? (int)uni(-1, 64) : (int)uni(0, 7);
  tally(stlt_forward(coin(40) ? nullptr : &m.p, coin(40) ? nullptr : &m.in, F<char>(120), ws, flags, coin(2) ? nullptr : FA<float>(121), F<float>(122), s));
  tally(stlt_backbone_forward(&m.p, &m.in, F<char>(120), ws, flags, F<float>(121), s));
  const size_t tape = coin(4) ? (size_t)uni(0, 1 << 24) : stlt_train_tape_bytes(m.in.B, m.in.T, m.in.N, m.p.d, m.p.n_spatial, m.p.n_temporal);
  const size_t scr = coin(4) ? (size_t)uni(0, 1 << 24) : stlt_train_scratch_bytes(m.in.B, m.in.T, m.in.N, m.p.d, m.p.n_categories);
  const int tflags = coin(3) ? (int)uni(0, 63) : (coin(2) ? 0 : 32);
  tally(stlt_train_forward(&m.p, &m.in, F<char>(123), tape, F<float>(124), coin(8) ? 1.f : 0.1f, rnd(), tflags, s));
  stlt_ctx* ctx = nullptr;
  if (coin(2)) (void)stlt_ctx_create(&ctx);
  Model g;
  g.sp = m.sp; g.tp = m.tp; g.p = m.p; g.p.spatial = g.sp.empty() ? nullptr : g.sp.data(); g.p.temporal = g.tp.empty() ? nullptr : g.tp.data();
  if (coin(3)) g.p.cat_emb = nullptr;  // a frozen parameter
  tally(stlt_train_backward(&m.p, coin(30) ? nullptr : &g.p, &m.in, F<char>(123), tape, F<char>(125), scr, F<float>(126), 0.1f, rnd(), tflags, ctx, s));
  // the fusion models' native call
  stlt_caf_params cp{};
  cp.layout = m.p;
  cp.feat_channels = coin(4) ? dim(2048, 1 << 16) : 2048;
  cp.app_tokens = coin(4) ? dim(32, 4096) : 32;
  cp.proj_w = FA<float>(130); cp.proj_b = FA<float>(131); cp.cls_token = FA<float>(132); cp.pos_embed = FA<float>(133);
  std::vector<stlt_layer_params> app((size_t)uni(0, 4), fake_layer(140));
  std::vector<stlt_crossmodal_params> fus((size_t)uni(0, 4));
  for (auto& c : fus) { const float** f = reinterpret_cast<const float**>(&c); for (size_t i = 0; i < sizeof(c) / sizeof(float*); ++i) f[i] = FA<float>(150 + (int)i); }
  cp.n_app_layers = coin(6) ? uni(-1, 9) : (int64_t)app.size();
  if ((size_t)(cp.n_app_layers > 0 ? cp.n_app_layers : 0) > app.size()) cp.n_app_layers = (int64_t)app.size();
  cp.app_layers = app.empty() ? nullptr : app.data();
  cp.n_fusion = (int64_t)fus.size();
  cp.fusion = fus.empty() ? nullptr : fus.data();
  { const float** f = reinterpret_cast<const float**>(&cp.fusion_head); for (int i = 0; i < 6; ++i) f[i] = FA<float>(180 + i); }
  if (coin(2)) { const float** f = reinterpret_cast<const float**>(&cp.layout_head); for (int i = 0; i < 12; ++i) f[i] = FA<float>(190 + i); }
  const size_t cws = coin(4) ? (size_t)uni(0, 1 << 24) : stlt_caf_workspace_bytes(m.in.B, m.in.T, m.in.N, m.p.d, cp.feat_channels, cp.app_tokens, m.p.n_classes);
  tally(stlt_caf_forward_flags(&cp, &m.in, F<float>(200), F<char>(201), cws, coin(2) ? 0 : 4, F<float>(202), coin(2) ? nullptr : FA<float>(203), coin(2) ? nullptr : FA<float>(204),
                               coin(2) ? nullptr : FA<float>(205), s));
  // block-level training calls and the context's own entry points
  const int64_t d = coin(5) ? hidden() : 768, H = d > 0 && d % 64 == 0 ? d / 64 : dim(12, 64), S = dim(2048, 1ll << 24), Lq = dim(33, 300), Lk = coin(2) ? Lq : dim(33, 300);
  stlt_attn_block_params ap; { const float** f = reinterpret_cast<const float**>(&ap); for (int i = 0; i < 6; ++i) f[i] = FA<float>(210 + i); }
  stlt_ffn_block_params fp; { const float** f = reinterpret_cast<const float**>(&fp); for (int i = 0; i < 6; ++i) f[i] = FA<float>(220 + i); }
  const bool cross = coin(2);
  tally(stlt_attn_block_fwd_train(&ap, d, H, 1e-12f, F<float>(230), Lq, cross ? FA<float>(231) : nullptr, Lk, F<uint8_t>(232), (int)uni(0, 1), S, 0.1f, rnd(), 0x400000, F<float>(233),
                                  cross ? FA<float>(234) : nullptr, F<float>(235), F<float>(236), F<float>(237), coin(2) ? nullptr : FA<char>(30), stlt_gemm_scratch_bytes(), s));
  const int64_t rows = S * (Lq > Lk ? Lq : Lk);
  const size_t kb = coin(4) ? (size_t)uni(0, 1 << 20) : stlt_block_keep_bytes(rows, d, 0), wb = coin(4) ? (size_t)uni(0, 1 << 20) : stlt_block_work_bytes(rows, d);
  if (ctx && coin(2)) (void)stlt_ctx_dw_defer(ctx, (int)uni(-2, 2));
  tally(stlt_attn_block_bwd_train(&ap, &ap, d, H, 1e-12f, F<float>(230), Lq, cross ? FA<float>(231) : nullptr, Lk, F<uint8_t>(232), (int)uni(0, 1), S, 0.1f, rnd(), 0x400000,
                                  F<float>(233), cross ? FA<float>(234) : nullptr, F<float>(235), F<float>(236), F<float>(238), F<float>(239), cross ? FA<float>(240) : nullptr, ctx,
                                  F<char>(241), kb, F<char>(242), wb, s));
  const int64_t M = dim(2112, 1ll << 28);
  const int act = coin(8) ? (int)uni(-1, 3) : (int)uni(1, 2);
  tally(stlt_ffn_block_fwd_train(&fp, d, 1e-5f, act, (int)uni(0, 1), F<float>(230), M, 0.1f, rnd(), 0x400000, act == 1 ? FA<float>(243) : nullptr, F<float>(244), F<float>(245),
                                 F<float>(246), coin(2) ? nullptr : FA<char>(30), stlt_gemm_scratch_bytes(), s));
  tally(stlt_ffn_block_bwd_train(&fp, &fp, d, 1e-5f, act, (int)uni(0, 1), F<float>(230), M, 0.1f, rnd(), 0x400000, act == 1 ? FA<float>(243) : nullptr, F<float>(244), F<float>(245),
                                 F<float>(247), F<float>(248), ctx, F<char>(241), coin(4) ? (size_t)uni(0, 1 << 20) : stlt_block_keep_bytes(M, d, 1), F<char>(242),
                                 coin(4) ? (size_t)uni(0, 1 << 20) : stlt_block_work_bytes(M, d), s));
  tally(stlt_linear_bwd(F<float>(1), F<float>(2), F<float>(3), dim(4096, 1ll << 31), dim(3072, 1 << 16), dim(3072, 1 << 16), coin(2) ? nullptr : FA<float>(4),
                        coin(2) ? nullptr : FA<float>(5), coin(2) ? nullptr : FA<float>(6), ctx, F<char>(7), coin(3) ? (size_t)uni(0, 1 << 26) : stlt_linear_bwd_scratch_bytes(3072), s));
  if (ctx) {
    stlt_wt_entry ent[40];
    const int64_t n = uni(-1, 40);
    for (int i = 0; i < 40; ++i) ent[i] = stlt_wt_entry{F<float>(250 + i), F<float>(300 + i), coin(8) ? dim(768, 1 << 24) : 4 * uni(1, 1024), coin(8) ? dim(768, 1 << 24) : 4 * uni(1, 1024)};
    tally(stlt_ctx_wt_refresh(ctx, coin(30) ? nullptr : ent, n, s));
    tally(stlt_input_grad_small(FA<float>(1), 768, ent[0].w, 768, 768, nullptr, 0, FA<float>(4), 768, dim(2048, 1 << 20), 0, ctx, s));
    (void)stlt_ctx_wt_hits(ctx);
    (void)stlt_ctx_dw_pending(ctx);
    tally(stlt_ctx_dw_flush(ctx, coin(3) ? nullptr : FA<char>(30), coin(4) ? 1024 : stlt_gemm_scratch_bytes(), s));
    tally(stlt_ctx_wt_clear(ctx));
    tally(stlt_ctx_destroy(ctx));
  }
  (void)stlt_set_train_side_stream((int)uni(-1, 1));
}

}  // namespace

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 2000;
  g_state = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
  if (stlt_version() != STLT_VERSION) { fprintf(stderr, "version mismatch\n"); return 2; }
  // handles that are not contexts are refused, not followed
  char junk[4096];
  memset(junk, 0x5a, sizeof junk);
  if (stlt_ctx_dw_defer(reinterpret_cast<stlt_ctx*>(junk), 1) == 0 || stlt_ctx_wt_clear(reinterpret_cast<stlt_ctx*>(junk)) == 0 || stlt_ctx_destroy(reinterpret_cast<stlt_ctx*>(junk)) == 0) {
    fprintf(stderr, "a junk context handle was accepted\n");
    return 3;
  }
  for (long i = 0; i < iters; ++i) {
    fuzz_bytes();
    fuzz_estimates();
    fuzz_products();
    fuzz_rowwise_and_attention();
    fuzz_whole_path();
    if ((i & 1023) == 1023) (void)stlt_prof_take_gemm_flops();
  }
  printf("host_fuzz: %ld iterations, %ld calls, %ld returned 0 (the rest refused their arguments); kernel launches configured: %ld , refused by the runtime: %ld ; last error: %s\n",
         iters, g_calls, g_ok, stlt_fake_hip_launches(), stlt_fake_hip_refused(), stlt_last_error());
  return 0;
}
