// Fused in-projection + attention core (SURVEY.md §8 row N1 / the north star's "fused MHSA"):
//   ctx[s, :, h] = softmax(q_h k_hᵀ / 8 + mask) v_h   with   [q_h | k_h | v_h] = X_s · W_in[h]ᵀ + b_in[h]
// for sequences of up to 64 tokens and 64-channel heads (nn.MultiheadAttention as configured at models.py:46-52,118-124; masks of
// models.py:68-71 (key padding) and utils/model_utils.py:4-7 + models.py:142-150 (causal + key padding)).  The packed QKV tensor
// never goes to HBM in inference: only X is read and ctx written.
//
// Round 3's kernel (32-token sequences only, 32x32 MFMA tiles, a wave = (clip, channel half), partial scores exchanged through
// LDS) is gone: the kernel below is faster on its own shape (64 / 256 / 1024 clips of 32 frames: 79.2 / 221.0 / 869.0 us against
// 85.7 / 242.2 / 896.1, profiles/round4_mhsa_fused_ab.txt) and takes every other one.
#include <cstdlib>
#include "common.h"
#include "wave_dpp.h"

namespace {

#ifndef STLT_MHSA_ABLATE
#define STLT_MHSA_ABLATE 0  // timing-only builds (wrong results): bit 0 no ctx stores (what keeping the attention output on chip for a fused out-projection could save at most); bit 2 no attention phase but its two barriers (the in-projection alone, accumulators kept alive): 5 749 against 6 077 us for 32 768 frames of 7 objects, 830 against 862 us for 1 024 clips of 32 frames (profiles/round6_mhsa_window_ab.txt)
#endif
#ifndef STLT_MHSA_SWAP_REDUCE
#define STLT_MHSA_SWAP_REDUCE 1  // 0: the softmax's two cross-group reductions through ds_bpermute shuffles (A/B builds)
#endif
#ifndef STLT_MHSA_EARLY_RESTART
#define STLT_MHSA_EARLY_RESTART 0  // 1 (A/B builds): the next item's bias rows and first fragments requested right behind attention barrier 2 (the q / k / v accumulators are dead there), under the softmax and PV, instead of after the phase — measured slower: 6 131 against 6 060 us (32 768 frames of 7), 866 - 877 against 860 us (1 024 clips of 32 frames), profiles/round6_mhsa_window_ab.txt
#endif
#ifndef STLT_MHSA_LOADER
#define STLT_MHSA_LOADER 2
#endif
constexpr int FM = 128, FN = 192, FK = 32;
constexpr int F_WAVES = 8, F_LOADERS = 4;
constexpr int F_THREADS = 64 * (F_WAVES + F_LOADERS);
constexpr int F_NSTAGE = 3;
constexpr int F_STAGE = (FM + FN) * FK;        // 10240 floats = 40 KB

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// Any sequence length up to 64 tokens (the reference's real layouts are T = layout_num_frames + 1 = 17 / 33, datasets.py:97-113;
// cfg4 has 64 frames), training forwards (probability dropout, packed QKV written for the tape) and — CAUSAL = false — the spatial
// tower's frames of N objects.
//
// Work item = (G = floor(128 / L) whole sequences = G·L <= 128 consecutive token rows, head): a 128 x 192 x d product (the 192
// output columns are the head's 64 q, k and v channels) on v_mfma_f32_16x16x4_f32, 8 MFMA waves + 4 DMA-only loader waves, with
// an MFMA wave owning ONE 16-row block and ALL 192 output columns (12 accumulators of
// 16 channels x 16 tokens, computed transposed: lane = token, registers = 4 consecutive channels).  That makes the attention
// phase independent of where sequences start inside the item:
//   * a wave's q accumulators are, as they stand, the B operand of Sᵀ = K·Qᵀ for its 16 queries over all 64 channels (no
//     partial-score exchange between waves);
//   * k and v leave the accumulators as 16-byte LDS stores into two 128-row tiles [token][channel] (chunk ^= token & 15): the K
//     tile lives in the operand stage the item's last k-step retired (the loaders refill it only after the phase's second
//     barrier), the V tile in 32 KB of its own;
//   * a query block needs the key blocks from the first row of its first query's sequence to its own block (causal) or to the
//     last row of its last query's sequence: NKB <= 5 blocks of 16 keys, all scores in registers (attn16.hip's one-pass softmax),
//     mask = (same sequence) & (key position <= query position) & (key not padded) from one metadata word per key row;
//   * Oᵀ = Vᵀ·Pᵀ with the probabilities in registers as the B operand; a lane stores 4 consecutive channels of its query per
//     channel block (16-byte stores), and in TRAIN builds its q / k / v accumulators to the tape's packed QKV rows.
// Operand staging is gemm.hip's: LDS-DMA with the source-side bank swizzle, three 40-KB stages, the loader waves two k-steps ahead
// with a counted vmcnt, one barrier per k-step, the bias strip DMA'd one item ahead; the 16-lane x 4-chunk fragment reads are
// conflict-free under that swizzle (every ds_read_b128 lane group covers all 16 slots of the 256-byte bank row).
constexpr int G_VT = FM * 64;                                             // V tile: 128 token rows x 64 channels
constexpr int G_SMEM = F_NSTAGE * F_STAGE + G_VT + 2 * FN + FM;           // + bias strips + key metadata: 39424 floats = 157696 B

struct Mhsa16Args {
  const float* X; const float* Win; const float* bin; const uint8_t* kpm;
  float* ctx; float* qkv;   // qkv: TRAIN only (packed rows [q;k;v], 3d floats per token)
  int n_tokens, L, rows_per_item, n_groups, H, d, head_sets;
  float scale;
  StltDrop dr; uint32_t site;
};

// WINDOW (round 6; non-causal launches whose sequences are not 16-row aligned, i.e. the spatial tower's frames of 5 - 8 objects): the key
// rows of a query block are read as 16-row blocks starting AT the first row of its first query's sequence instead of at the 16-row block
// holding that row.  The keys a block of 16 queries of 7-token frames can see span at most 28 rows: 2 blocks from the sequence's first row,
// 3 from the aligned block (20 -> 15 key blocks per 126-row item, NKB 3 -> 2).  The LDS tiles are swizzled by (row & 15) and a block is 16
// consecutive rows wherever it starts, so the reads stay conflict-free; the launcher takes this form only when every block of every window
// lies inside the 128-row tile (mhsa16_window).
template <int NKB, bool CAUSAL, bool TRAIN, bool WINDOW>
__global__ __launch_bounds__(F_THREADS, 3) void mhsa16_kernel(const Mhsa16Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d, H = a.H;
  const int nk = d / FK;
  const int M = a.n_tokens;
  const int n_items = a.n_groups * H;
  const int G = gridDim.x;
  // Item order.  head_sets == 1: items head-fastest over XCD-contiguous virtual workgroup ids — an XCD's 32 workgroups then hold
  // ~3 row groups x all H heads at a time: every X tile is fetched about once, but the H weight slices (3 d^2 floats = 7 MB at d = 768)
  // do not fit the 4-MB L2 beside the X stream and are re-fetched every round (measured 4.35x the algorithmic bytes, round 3).
  // head_sets == 4 (H % 4 == 0, full grid): XCD x works on head set x & 3 (H / 4 heads: 1.8 MB of weights, L2-resident for the
  // whole launch) for one half of the row groups (x >> 2); an X tile is then fetched once per head set: ~4x X, ~1x W.
  const int sets = a.head_sets;
  int v = blockIdx.x, Gv = G, hps = H, head0 = 0, grp0 = 0, my_groups = a.n_groups;
  if (sets > 1) {
    const int x = blockIdx.x & 7;
    hps = H / sets;
    head0 = (x % sets) * hps;
    const int halves = 8 / sets, half = x / sets;         // XCDs sharing a head set split the row groups
    const int per = (a.n_groups + halves - 1) / halves;
    grp0 = half * per;
    my_groups = a.n_groups - grp0 < per ? a.n_groups - grp0 : per;
    if (my_groups < 0) my_groups = 0;
    v = blockIdx.x >> 3;
    Gv = G >> 3;
  } else if ((G & 7) == 0) {
    v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);  // XCD-contiguous virtual id (round-robin dispatch)
  }
  const int my_n = sets > 1 ? my_groups * hps : n_items;
  const int my_items = (my_n - v + Gv - 1) / Gv;
  if (my_items <= 0) return;
  const int total_steps = my_items * nk;
  float* Vt = smem + F_NSTAGE * F_STAGE;
  float* bias_lds = Vt + G_VT;
  int* kmeta = reinterpret_cast<int*>(bias_lds + 2 * FN);
  auto item_of = [&](int it, int& grp, int& head) {
    const int item = v + it * Gv;
    const int gq = item / hps;
    grp = grp0 + gq;
    head = head0 + item - gq * hps;
  };

  if (wave >= F_WAVES) {
    // ---- loader waves: loader Ld issues A rows [32 Ld, 32 Ld + 32) and B rows [48 Ld, 48 Ld + 48); B image row r = (q|k|v = r / 64, channel r % 64)
    const int Ld = wave - F_WAVES;
    const int drow = lane >> 3, dslot = lane & 7;
    // addresses = a wave-uniform base (scalar registers, advanced per k-step by scalar adds) + a per-lane 32-bit byte offset fixed for
    // the item: no vector-ALU instruction in the steady state (see gemm16_kernel.h).  An instruction's 8 rows of the W image lie in one
    // of the head's q / k / v blocks (8 divides 64), so that block's first row goes into the instruction's base.
    // STLT_MHSA_LOADER (A/B builds): 0 = round 4's 64-bit per-lane pointers + a vector add per instruction, 1 = the compiler's builtin on base + offset
    // (it folds the X image's into the scalar-base form and keeps a vector add for the W image's), 2 = stlt_dma16 for every instruction
    uint32_t voa[4], vob[6];
    const char* bx = nullptr;
    const char* bw = nullptr;
#if STLT_MHSA_LOADER == 0
    const float* pa[4];
    const float* pb[6];
#endif
    auto set_item = [&](int it) {
      int grp, head;
      item_of(it, grp, head);
      const int row0 = grp * a.rows_per_item;
      bx = reinterpret_cast<const char*>(a.X + (int64_t)row0 * d);
      bw = reinterpret_cast<const char*>(a.Win + (int64_t)head * 64 * d);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = Ld * 32 + i * 8 + drow;
        const int rr = r < M - row0 ? r : M - 1 - row0;  // rows past the batch re-read the last row; nothing of theirs is stored
        voa[i] = ((uint32_t)rr * (uint32_t)d + (uint32_t)((dslot ^ ((r >> 1) & 7)) * 4)) * 4u;
#if STLT_MHSA_LOADER == 0
        pa[i] = a.X + (int64_t)(row0 + rr) * d + (dslot ^ ((r >> 1) & 7)) * 4;
#endif
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int r = Ld * 48 + i * 8 + drow;
        vob[i] = ((uint32_t)(r & 63) * (uint32_t)d + (uint32_t)((dslot ^ ((r >> 1) & 7)) * 4)) * 4u;
#if STLT_MHSA_LOADER == 0
        pb[i] = a.Win + (int64_t)((r >> 6) * d + head * 64 + (r & 63)) * d + (dslot ^ ((r >> 1) & 7)) * 4;
#endif
      }
    };
    auto dma_bias = [&](int it) {
      if (Ld == 0) {
        int grp, head;
        item_of(it, grp, head);
        float* dst = bias_lds + (it & 1) * FN;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (STLT_MHSA_LOADER == 2) stlt_dma4(a.bin + i * d + head * 64, (uint32_t)lane * 4u, stlt_lds_addr(dst + i * 64));
          else __builtin_amdgcn_global_load_lds((glb_void_ptr)(a.bin + i * d + head * 64 + lane), (lds_void_ptr)(dst + i * 64), 4, 0, 0);
        }
      }
    };
    int l_it = 0, l_kt = 0, l_stage = 0;
    auto l_step = [&]() {
      if (l_kt == 0) set_item(l_it);
      float* sa = smem + l_stage * F_STAGE + (Ld * 32) * FK;
      float* sb = smem + l_stage * F_STAGE + FM * FK + (Ld * 48) * FK;
#if STLT_MHSA_LOADER == 0
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_void_ptr)(pa[i] + l_kt * FK), (lds_void_ptr)(sa + i * 8 * FK), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) __builtin_amdgcn_global_load_lds((glb_void_ptr)(pb[i] + l_kt * FK), (lds_void_ptr)(sb + i * 8 * FK), 16, 0, 0);
#else
      const uint32_t la = stlt_lds_addr(sa), lb = stlt_lds_addr(sb);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (STLT_MHSA_LOADER == 1) __builtin_amdgcn_global_load_lds((glb_void_ptr)(bx + voa[i]), (lds_void_ptr)(sa + i * 8 * FK), 16, 0, 0);
        else stlt_dma16(bx, voa[i], la + i * 8 * FK * 4);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int qkv = (Ld * 48 + i * 8) >> 6;  // wave-uniform: the q / k / v block of this instruction's 8 rows
        const char* bwi = bw + (int64_t)qkv * d * d * (int64_t)sizeof(float);
        if (STLT_MHSA_LOADER == 1) __builtin_amdgcn_global_load_lds((glb_void_ptr)(bwi + vob[i]), (lds_void_ptr)(sb + i * 8 * FK), 16, 0, 0);
        else stlt_dma16(bwi, vob[i], lb + i * 8 * FK * 4);
      }
      bx += FK * sizeof(float);
      bw += FK * sizeof(float);
#endif
      if (++l_kt == nk) { ++l_it; l_kt = 0; }
      if (++l_stage == F_NSTAGE) l_stage = 0;
    };
    dma_bias(0);
    l_step();
    if (total_steps > 1) {
      l_step();
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // in-order counter: step 0 and the bias strip landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int w_it = 0, w_kt = 0;  // position of the MFMA waves
    for (int step = 0; step < total_steps; ++step) {
      if (w_kt == nk - 1 && w_it + 1 < my_items) dma_bias(w_it + 1);
      if (step + 2 < total_steps) {
        l_step();
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // step+1 landed; only step+2's ten instructions may stay in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (++w_kt == nk) {
        ++w_it;
        w_kt = 0;
        __builtin_amdgcn_s_barrier();  // the MFMA waves' attention phase: K / V tiles written ...
        __builtin_amdgcn_s_barrier();  // ... and the K tile (= the stage the next l_step refills) read
      }
    }
    return;
  }

  // ---- MFMA waves.  Row block of the wave: SIMD partners (waves w and w + 4) get blocks b and 7 - b, whose causal key-block
  // counts add up to the same number on every SIMD when sequences are block-aligned (L = 32: 1 + 2, L = 64: 1 + 4, 2 + 3)
  const int rb = wave < 4 ? wave : 11 - wave;
  const int li_ = lane & 15, lg_ = lane >> 4;
  const int sw = (li_ >> 1) & 7;  // rows 16 rb + li_ and 16 t + li_ share (row >> 1) & 7
  const int x_row = (rb * 16 + li_) * FK;
  const int w_row = (FM + li_) * FK;  // + 16 * FK per 16-channel tile
  // A k-step is four phases (k-chunk c = p >> 1 of 16, column tiles 6 (p & 1) .. + 5); the chunk's X fragment is read once and shared by
  // its two halves; the fragments of phase p + 1 are requested behind the first MFMA group of phase p (gemm16_kernel.h has the why)
  struct Frags { f32x4 w[6]; };
  f32x4 xf[2];
  Frags F[2];
  auto read_phase = [&](int stage, int p, Frags& f) {
    const float* s = smem + stage * F_STAGE;
    const int c = p >> 1;
    const int off = ((4 * c + lg_) ^ sw) * 4;
    if ((p & 1) == 0) xf[c] = *reinterpret_cast<const f32x4*>(s + x_row + off);
#pragma unroll
    for (int t = 0; t < 6; ++t) f.w[t] = *reinterpret_cast<const f32x4*>(s + w_row + (6 * (p & 1) + t) * 16 * FK + off);
  };
  f32x4 acc[12];  // tile t = 16 output columns: q channels 16 t .. (t < 4), k (4 <= t < 8), v (t >= 8); lane (li_, lg_): token li_, channels 4 lg_ .. + 3
  auto init_acc = [&](int it) {
    const float* src = bias_lds + (it & 1) * FN + 4 * lg_;
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = *reinterpret_cast<const f32x4*>(src + 16 * t);
  };
  auto mfma_phase = [&](int p, const Frags& f, int e_lo, int e_hi) {
    const int c = p >> 1, t0 = 6 * (p & 1);
#pragma unroll
    for (int e = e_lo; e < e_hi; ++e)
#pragma unroll
      for (int t = 0; t < 6; ++t)
        acc[t0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[t][e], xf[c][e], acc[t0 + t], 0, 0, 0);
  };

  __builtin_amdgcn_s_barrier();  // the loaders' counted wait + this barrier publish step 0 and the first bias strip
  init_acc(0);
  int c_it = 0, c_kt = 0, stage = 0;
  unsigned pad_byte = 1u;
  read_phase(0, 0, F[0]);
  for (int step = 0; step < total_steps; ++step) {
    const int next_stage = stage + 1 == F_NSTAGE ? 0 : stage + 1;
    if (c_kt == 0) {  // this item's key-padding byte of the wave's row li_: in flight under the whole product
      int grp, head;
      item_of(c_it, grp, head);
      const int g_row = grp * a.rows_per_item + rb * 16 + li_;
      pad_byte = (rb * 16 + li_ < a.rows_per_item && g_row < M) ? (unsigned)a.kpm[g_row] : 1u;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      mfma_phase(p, F[p & 1], 0, 1);  // in front of the next phase's requests: the compiler's wait for this phase's fragments is an lgkmcnt(0)
      read_phase(stage, p + 1, F[(p + 1) & 1]);
      mfma_phase(p, F[p & 1], 1, 4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own fragment reads of this stage are done
    __builtin_amdgcn_s_barrier();                        // retire the stage; step+1 landed
    const bool item_done = c_kt + 1 == nk;
    if (!item_done) read_phase(next_stage, 0, F[0]);  // (at the end of an item the attention phase comes first: fewer live registers there)
    mfma_phase(3, F[1], 0, 4);
    const int dead_stage = stage;
    stage = next_stage;
    ++c_kt;
    if (!item_done) continue;

    // ---- attention phase of item c_it
    int grp, head;
    item_of(c_it, grp, head);
    // opaque copies of the lane coordinates: the phase's address arithmetic is recomputed per item (a few dozen VALU) instead of
    // being hoisted out of the item loop into registers the k-loop cannot spare (it spilled 12 - 51 of them)
    int li = li_, lg = lg_;
    asm volatile("" : "+v"(li), "+v"(lg));
    const int row0 = grp * a.rows_per_item;                                         // first token row of the item
    const int rows_here = (M - row0) < a.rows_per_item ? (M - row0) : a.rows_per_item;  // whole sequences: a multiple of L
    const int L = a.L;
    float* Kt = smem + dead_stage * F_STAGE;
    const int my_row = rb * 16 + li;
    const bool row_ok = my_row < rows_here;
    const int q_seq = my_row / L, q_pos = my_row - q_seq * L;
    if (!(STLT_MHSA_ABLATE & 4)) {  // k, v -> LDS tiles; this row's key metadata: -1 = absent / padded, else (sequence in item << 8) | position
      const int base = my_row * 64;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const int off = base + (((cb * 4 + lg) ^ li) * 4);
        *reinterpret_cast<f32x4*>(Kt + off) = acc[4 + cb];
        *reinterpret_cast<f32x4*>(Vt + off) = acc[8 + cb];
      }
      if (lg == 0) kmeta[my_row] = (row_ok && pad_byte == 0u) ? ((q_seq << 8) | q_pos) : -1;
      if (TRAIN && row_ok) {
        float* qrow = a.qkv + (int64_t)(row0 + my_row) * (3 * d) + head * 64 + 4 * lg;
#pragma unroll
        for (int t = 0; t < 12; ++t) *reinterpret_cast<f32x4*>(qrow + (t >> 2) * d + (t & 3) * 16) = acc[t];
      }
    }
    // key rows of this query block: from the first row of its first query's sequence (WINDOW) or the 16-row block holding it ...
    const int blk_row0 = rb * 16;
    const bool blk_ok = blk_row0 < rows_here;  // wave-uniform
    const int seq_row0 = (blk_row0 / L) * L;
    const int k0 = WINDOW ? seq_row0 : (seq_row0 & ~15);
    int last_row = blk_row0 + 15;  // ... to its own block (causal), or to the last row of its last query's sequence
    if (!CAUSAL) {
      const int r_last = blk_row0 + 15 < rows_here ? blk_row0 + 15 : rows_here - 1;
      last_row = (r_last / L + 1) * L - 1;
    }
    const int n_kb = ((last_row - k0) >> 4) + 1;  // key blocks i = 0 .. n_kb - 1 hold rows k0 + 16 i .. + 15
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // attention barrier 1: every wave's K / V rows and metadata are in LDS
    f32x4 st[NKB];
#pragma unroll
    for (int i = 0; i < NKB; ++i) {
      st[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (blk_ok && i < n_kb && !(STLT_MHSA_ABLATE & 4)) {
        const int krow_i = k0 + 16 * i + li;  // WINDOW: inside the tile's 128 rows (mhsa16_window checks the shape)
        const float* krow = Kt + krow_i * 64;
        const int ksw = WINDOW ? (krow_i & 15) : li;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + (((cb * 4 + lg) ^ ksw) * 4));
#pragma unroll
          for (int r = 0; r < 4; ++r) st[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[r], acc[cb][r], st[i], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // attention barrier 2: the K tile is read (the loaders may refill its stage); q, k, v accumulators are dead
    if (STLT_MHSA_EARLY_RESTART && step + 1 < total_steps) {  // next item: accumulators from its bias strip (published by the last k-step's barrier), first fragments
      init_acc(c_it + 1);
      read_phase(stage, 0, F[0]);
    }
    if (STLT_MHSA_ABLATE & 4) {
#pragma unroll
      for (int t = 0; t < 12; ++t) asm volatile("" :: "v"(acc[t]));  // the product is kept
    } else if (blk_ok) {
      // mask + softmax: st[i][r] = score of key k0 + 16 i + 4 lg + r against query li
      float m = -1e30f;
#pragma unroll
      for (int i = 0; i < NKB; ++i) {
        if (i < n_kb) {
          const int j0 = k0 + 16 * i + 4 * lg;
          int meta[4];
          if (WINDOW) {  // 4-byte aligned only
#pragma unroll
            for (int r = 0; r < 4; ++r) meta[r] = kmeta[j0 + r];
          } else {
            const int4 km = *reinterpret_cast<const int4*>(kmeta + j0);
            meta[0] = km.x; meta[1] = km.y; meta[2] = km.z; meta[3] = km.w;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = (meta[r] >= 0) & ((meta[r] >> 8) == q_seq) & (!CAUSAL || (meta[r] & 0xff) <= q_pos);
            st[i][r] = ok ? st[i][r] * a.scale : -1e30f;
            m = fmaxf(m, st[i][r]);
          }
        }
      }
      if (STLT_MHSA_SWAP_REDUCE) {  // over the four 16-lane groups: permlane swaps, no LDS round trips on the phase's critical path
        m = groups_max(m);
      } else {
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
      }
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NKB; ++i) {
        if (i < n_kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = st[i][r] > -1e29f ? __expf(st[i][r] - m) : 0.f;
            st[i][r] = p;
            sum += p;
          }
        }
      }
      if (STLT_MHSA_SWAP_REDUCE) {
        sum = groups_sum(sum);
      } else {
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
      }
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;  // fully masked row -> zeros
      // TRAIN: dropout of the probabilities, applied where a probability is consumed (attn.hip's element index:
      // ((query token * H + head) << 8) | key position; an unmasked key is in the query's sequence, so its position is its row minus
      // the sequence's first row; masked keys have probability 0 either way)
      const uint64_t drop_key = TRAIN ? stlt_drop_key(a.dr, a.site) : 0ull;
      const uint64_t qidx = (((uint64_t)(row0 + my_row)) * H + head) << 8;
      // Oᵀ[channel][query] += Vᵀ·Pᵀ: MFMA step (i, r) sums keys k0 + 16 i + 4 g + r over g
      f32x4 o[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) o[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NKB; ++i) {
        if (i < n_kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = k0 + 16 * i + 4 * lg + r;
            float pr = st[i][r];
            if (TRAIN && a.dr.thr) pr = stlt_keep_k(a.dr.thr, drop_key, qidx | (uint64_t)((j - q_seq * L) & 0xff)) ? pr * a.dr.scale : 0.f;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
              const float vv = Vt[j * 64 + (((cb * 4 + (li >> 2)) ^ (j & 15)) * 4) + (li & 3)];
              o[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, pr, o[cb], 0, 0, 0);
            }
          }
        }
      }
      if (row_ok) {
        float* dst = a.ctx + (int64_t)(row0 + my_row) * d + head * 64 + 4 * lg;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          const f32x4 val = o[cb] * inv;
          if (STLT_MHSA_ABLATE & 1) asm volatile("" :: "v"(val), "v"(dst));  // timing build: the attention output is computed, not stored
          else *reinterpret_cast<f32x4*>(dst + 16 * cb) = val;
        }
      }
    }
    ++c_it;
    c_kt = 0;
    if (!STLT_MHSA_EARLY_RESTART && step + 1 < total_steps) {
      init_acc(c_it);
      read_phase(stage, 0, F[0]);
    }
  }
}

// largest number of 16-key blocks a 16-query block of an item needs (host side; the kernel's NKB); window: blocks counted from the first
// row of the block's first sequence instead of from the 16-row block holding it
static int mhsa16_key_blocks_as(int L, int rows_per_item, bool causal, bool window, int* total = nullptr, bool* all_inside = nullptr) {
  int worst = 1, sum = 0;
  bool inside = true;  // every key block within the item's 128-row K / V tiles
  for (int b = 0; b * 16 < rows_per_item; ++b) {
    const int seq_row0 = (b * 16 / L) * L;
    const int k0 = window ? seq_row0 : (seq_row0 & ~15);
    int last_row = b * 16 + 15;
    if (!causal) {
      const int r_last = b * 16 + 15 < rows_per_item ? b * 16 + 15 : rows_per_item - 1;
      last_row = (r_last / L + 1) * L - 1;
    }
    const int n = ((last_row - k0) >> 4) + 1;
    if (n > worst) worst = n;
    sum += n;
    if (k0 + 16 * n > FM) inside = false;
  }
  if (total) *total = sum;
  if (all_inside) *all_inside = inside;
  return worst;
}
// STLT_MHSA_WINDOW=0: aligned key blocks everywhere (A/B runs); default: the window form where an item needs fewer key blocks with it
// (non-causal only; the window never needs more per query block)
static bool mhsa16_window(int L, int rows_per_item, bool causal) {
  static const int on = [] { const char* e = getenv("STLT_MHSA_WINDOW"); return e ? atoi(e) : 1; }();
  if (!on || causal) return false;
  int total_w = 0, total_a = 0;
  bool inside = false;
  const int worst_w = mhsa16_key_blocks_as(L, rows_per_item, false, true, &total_w, &inside), worst_a = mhsa16_key_blocks_as(L, rows_per_item, false, false, &total_a);
  return inside && worst_w <= worst_a && total_w < total_a;
}
static int mhsa16_key_blocks(int L, int rows_per_item, bool causal) {
  return mhsa16_key_blocks_as(L, rows_per_item, causal, mhsa16_window(L, rows_per_item, causal));
}

template <int NKB, bool CAUSAL, bool TRAIN, bool WINDOW>
static int launch_mhsa16_as(const Mhsa16Args& a, hipStream_t s) {
  static StltPerDeviceOnce attr_done;
  if (!attr_done.flag()) {
    if (hipError_t e = hipFuncSetAttribute((const void*)mhsa16_kernel<NKB, CAUSAL, TRAIN, WINDOW>, hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM * (int)sizeof(float)); e != hipSuccess)
      return stlt_set_error((int)e, "mhsa16: %s", hipGetErrorString(e));
    attr_done.flag() = true;
  }
  const int64_t n_items = (int64_t)a.n_groups * a.H;
  int64_t G = stlt_device_cus();
  if (G > n_items) G = n_items;
  Mhsa16Args b = a;
  // head sets per XCD group (see the kernel): only for launches of several rounds on a full grid of 8 x n workgroups.  Measured at
  // 1024 clips of 32 frames (profiles/round4_mhsa_head_sets.txt): memory-side traffic per launch 907 MB (1 set) -> 668 MB (2 sets) ->
  // 699 MB (4 sets) at equal speed (930 / 926 / 930 us) -> 2 sets.  The spatial launches (7 objects, 85 rounds of items) keep 1 set: with
  // 2 sets their traffic went UP, 9.4 -> 11.5 GB per launch (1.4 GB algorithmic: six heads' weight slices still do not fit the L2 beside
  // the X stream, and X is then fetched twice), and the launch from 6276 to 6376 us; 4 sets: 6715 us.  The re-fetched bytes are the 7 MB
  // of weights, served by the Infinity Cache (9.4 GB in 6.3 ms = 1.5 TB/s); the kernel stays MFMA-bound at 0.83.
  // STLT_MHSA_HEAD_SETS=1|2|4 forces a value for both towers (A/B runs).
  static const int env_sets = [] { const char* e = getenv("STLT_MHSA_HEAD_SETS"); return e ? atoi(e) : 0; }();
  const int want_sets = env_sets ? env_sets : (CAUSAL ? 2 : 1);
  b.head_sets = 1;
  if ((want_sets == 2 || want_sets == 4) && (G & 7) == 0 && a.H % want_sets == 0 && n_items >= 4 * G && a.n_groups >= 8) b.head_sets = want_sets;
  hipLaunchKernelGGL((mhsa16_kernel<NKB, CAUSAL, TRAIN, WINDOW>), dim3((unsigned)G), dim3(F_THREADS), G_SMEM * sizeof(float), s, b);
  return stlt_check_launch("mhsa16_kernel");
}

template <bool CAUSAL, bool TRAIN, bool WINDOW>
static int launch_mhsa16_nkb(int nkb, const Mhsa16Args& a, hipStream_t s) {
  switch (nkb) {
    case 1: return launch_mhsa16_as<1, CAUSAL, TRAIN, WINDOW>(a, s);
    case 2: return launch_mhsa16_as<2, CAUSAL, TRAIN, WINDOW>(a, s);
    case 3: return launch_mhsa16_as<3, CAUSAL, TRAIN, WINDOW>(a, s);
    case 4: return launch_mhsa16_as<4, CAUSAL, TRAIN, WINDOW>(a, s);
    default: return launch_mhsa16_as<5, CAUSAL, TRAIN, WINDOW>(a, s);
  }
}

}  // namespace

// Does the fused kernel take this shape?  Sequences of 1..64 tokens, 64-channel heads, d a multiple of the 32-wide k-step, and at
// most 5 key blocks per query block (always true for causal sequences; non-causal: up to ~36 tokens).
bool stlt_mhsa_fused_takes(int64_t L, int64_t H, int64_t d, int causal) {
  if (L < 1 || L > 64 || H <= 0 || H > 256 || d != H * 64 || d % FK != 0) return false;  // (H <= 256: the loaders' 32-bit weight offsets)
  const int rows = (int)(FM / L * L);
  return mhsa16_key_blocks((int)L, rows, causal != 0) <= 5;
}

// Is the fused launch the faster form for S sequences?  Launch-time estimates fitted to stand-alone measurements on MI355X
// (profiles/round5_mhsa_fused_ab.txt, round5_gemm16_shapes.txt):
//   fused  = rounds of items x (k-steps x 2.89 us + 0.7 us per key block + 0.6) + 6 us — whole sequences per 128-row item, so
//            33 tokens fill 99 rows and pay for 128 (1219 us against the pair's 981 at 1024 clips), and a launch pays for whole
//            rounds of workgroups (64 frames x 64 clips: 1.5 rounds = 2);
//   pair   = the in-projection as launch_linear would run it (large tiles, stream-K, or gemm16's small tiles) + the attention core
//            at ~5.2 TB/s of its 16 bytes per token and channel + 6 us.
// Fused wins from ~256 clips on for 17 / 32 / 64 frames (854 against 917 us at 1024 clips of 32 frames, 434 against 460 at 256 clips of 64)
// and for 5 - 8 object slots at bench sizes (6180 against 6251 us at 32768 frames of 7); the pair wins at 64 clips (small tiles: 65 + 10 us
// against 79), for 33 frames and for 36 objects.  STLT_FUSED_MHSA_FORCE=1: whenever the shape is taken (A/B runs).
bool stlt_mhsa_fused_pays(int64_t S, int64_t L, int64_t H, int64_t d, int causal) {
  if (!stlt_mhsa_fused_takes(L, H, d, causal)) return false;
  static const int force = [] { const char* e = getenv("STLT_FUSED_MHSA_FORCE"); return e ? atoi(e) : 0; }();
  if (force) return true;
  const int64_t seq_per_item = FM / L;
  const int64_t n_items = (S + seq_per_item - 1) / seq_per_item * H;
  const int64_t cus = stlt_device_cus();
  const int64_t rounds = (n_items + cus - 1) / cus;
  const int nkb = mhsa16_key_blocks((int)L, (int)(seq_per_item * L), causal != 0);
  const double fused = (double)rounds * ((double)(d / FK) * 2.89 + 0.7 * nkb + 0.6) + 6.0;
  const double pair = stlt_linear_est_us(S * L, 3 * d, d) + 6.0 + 16.0 * (double)(S * L) * (double)d / 5.2e6;
  return fused < pair;
}

// ctx (S*L, d) = multi-head self-attention of S sequences of L tokens with the in-projection fused in.
// x (S*L, d), w_in (3d, d) rows [q; k; v], b_in (3d), kpm (S*L) bytes (1 = padded key); causal: key position <= query position.
// qkv_out != nullptr: the packed projections (S*L, 3d) are also written (training tape); dr: dropout of the probabilities.
int launch_mhsa_fused(const float* x, const float* w_in, const float* b_in, const uint8_t* kpm, int64_t S, int64_t L, int64_t H,
                      int64_t d, float* ctx, hipStream_t s, int causal, float* qkv_out, StltDrop dr, uint32_t site) {
  if (!x || !w_in || !b_in || !kpm || !ctx) return stlt_set_error(STLT_EINVAL, "mhsa_fused: null pointer");
  if (!stlt_mhsa_fused_takes(L, H, d, causal))
    return stlt_set_error(STLT_EINVAL, "mhsa_fused: sequences of 1..64 tokens and 64-channel heads (L=%lld, d=%lld, H=%lld, causal=%d)", (long long)L,
                          (long long)d, (long long)H, causal);
  if (S <= 0) return 0;
  if (S * L * 3 * d > 0x7fffffffLL * 4 || S * L > 0x7fffff00LL || (S / (FM / L) + 1) * H > 0x3fffffffLL)
    return stlt_set_error(STLT_EINVAL, "mhsa_fused: batch too large");
  StltProfScope ps(causal ? STLT_K_MHSA_FUSED : STLT_K_MHSA_FUSED_SPATIAL, s);
  Mhsa16Args a;
  a.X = x; a.Win = w_in; a.bin = b_in; a.kpm = kpm; a.ctx = ctx; a.qkv = qkv_out;
  a.n_tokens = (int)(S * L); a.L = (int)L; a.H = (int)H; a.d = (int)d;
  const int seq_per_item = (int)(FM / L);
  a.rows_per_item = seq_per_item * (int)L;
  a.n_groups = (int)((S + seq_per_item - 1) / seq_per_item);
  a.scale = 0.125f;  // 1 / sqrt(64)
  a.dr = dr; a.site = site;
  const int nkb = mhsa16_key_blocks(a.L, a.rows_per_item, causal != 0);
  const bool train = qkv_out != nullptr || dr.thr != 0;
  {
    const long long items = (long long)a.n_groups * H, cus = stlt_device_cus();
    stlt_prof_note("mhsa16 S=%lld L=%lld H=%lld d=%lld causal=%d train=%d item=%dx192 items=%lld wg=%lld rounds=%lld ksteps=%lld keyblocks=%d%s", (long long)S, (long long)L,
                   (long long)H, (long long)d, causal, (int)train, a.rows_per_item, items, items < cus ? items : cus, (items + cus - 1) / cus, (long long)(d / 32), nkb, mhsa16_window(a.L, a.rows_per_item, causal != 0) ? "w" : "");
    stlt_prof_note_flops(6.0 * (double)(S * L) * (double)d * (double)d + 4.0 * (double)S * (double)H * (double)(L * L) * 64.0);
    stlt_prof_add_bytes(8.0 * (double)(S * L) * (double)d + 4.0 * (3.0 * d * d + 3.0 * d) + (double)(S * L));
  }
  if (train && !qkv_out) return stlt_set_error(STLT_EINVAL, "mhsa_fused: dropout needs the qkv output (training forward)");
  if (causal) return train ? launch_mhsa16_nkb<true, true, false>(nkb, a, s) : launch_mhsa16_nkb<true, false, false>(nkb, a, s);
  if (mhsa16_window(a.L, a.rows_per_item, false)) return train ? launch_mhsa16_nkb<false, true, true>(nkb, a, s) : launch_mhsa16_nkb<false, false, true>(nkb, a, s);
  return train ? launch_mhsa16_nkb<false, true, false>(nkb, a, s) : launch_mhsa16_nkb<false, false, false>(nkb, a, s);
}
