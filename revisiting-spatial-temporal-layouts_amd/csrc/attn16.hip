// K3, 16-row tiles: the attention core for self-attention over short sequences (L <= 64 tokens on a packed QKV buffer; DROP builds:
// with the train-mode dropout of the probabilities) on v_mfma_f32_16x16x4_f32.  attn.hip's 32x32 tiles are exact for L = 32 / 64 and for 4 x 7-token frames, but
// a 36-token sequence (cfg4: N = 36) costs them four (query tile, key tile) steps of which 68 % is padding, and a wave
// needs 64 dependent 64-cycle MFMAs per (sequence, head).  Here an item is up to four 16-row blocks:
//   FULL  (16 < L <= 64): one sequence = NB = ceil(L/16) blocks; every (query block, key block) pair, or the lower
//         triangle when causal — cfg2 temporal (T = 32): 3 pairs instead of a 32x32 tile; cfg4 spatial (N = 36): 9 pairs of
//         16x16 instead of 4 of 32x32; cfg4 temporal (T = 64): 10 pairs;
//   DIAG  (L <= 16): P = floor(16/L) whole sequences per block, NB independent blocks per item, diagonal pairs only —
//         cfg2 spatial (N = 7): 2 frames per block.
// Dataflow: S^T = K·Q^T with swapped operands (a query's scores sit in 4 lanes x 4 registers per key block: row max / sum
// are in-register + two shuffles, and the probabilities are already the B operand of O^T += V^T·P^T), Q and K fragments
// straight from global memory in operand shape, V by LDS-DMA into a swizzled tile.  A wave works through an item one
// query block at a time — its Q·K^T blocks, one-pass softmax (all of a query's scores are in registers), P·V, store —
// holding the K fragments of every block, one query block's Q fragments, NB score and 4 output accumulators: half the
// registers and LDS of a schedule that computes every score block first (measured: 8 instead of 6 waves per CU at three
// blocks, cfg4 spatial 524 -> 403 us).  The next query block's Q fragments are fetched while this one's softmax runs; the
// next item's K / Q fragments as soon as the last Q·K^T has consumed the registers, its V tile as soon as the last P·V
// has read the (single) V buffer.  The output leaves without an LDS transpose: lane (query, g) holds 4 consecutive
// channels per channel block = one 16-byte store.
// TAIL  (causal, L = 16 NB + 1 — the reference's real layouts: T = layout_num_frames + 1 = 17 / 33, datasets.py:97-113): the last token
//         is seen as a key by itself only and as a query sees every key, so the item is NB full blocks plus one row handled beside them:
//         its scores against the NB key blocks come from one more Q·K^T pass (the query block holds 16 copies of that row), its own
//         key's score and value row are a 64-channel dot product and an axpy on the vector ALU.  No (NB + 1)-th key block is held, no
//         V rows beyond 16 NB sit in LDS: the register / LDS footprint of the NB-block kernel instead of the NB + 1 one (33 tokens as three
//         blocks: 208 VGPRs, two waves per SIMD, 0.57 of the HBM peak).
// SPLIT (small grids): the unit of work is (item, query block) instead of item, so that a launch with fewer items than
// wave slots spreads over NB times as many waves and each wave's dependent chain is one query block long; a unit loads
// only the key / value blocks it uses.
// Bit-for-bit it is a different summation order than attn.hip (16-key blocks); both are tested against the same fp64 oracle.
#include <cstdlib>
#include "common.h"
#include "wave_dpp.h"

namespace {

constexpr int DH16 = 64;
constexpr int WAVES16 = 4;  // independent waves per workgroup

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

struct Geo16 {
  const float* qkv;      // packed (tokens, 3*H*64): q | k | v
  const uint8_t* kpm;    // one byte per token: 1 = padded key
  float* ctx;            // (tokens, H*64)
  int n_tokens, L, H, P; // P = sequences per block (DIAG), 1 for FULL
  int rows_per_item;     // FULL: L; DIAG: NB * P * L
  int n_items, reverse;
  float scale;
  StltDrop dr;           // DROP builds: train-mode dropout of the probabilities (attn.hip's mask: ((query token * H + head) << 8) | key position)
  uint32_t site;
};

// (TAIL with two full blocks — 33 tokens — is held to three waves per SIMD: the compiler's own allocation is 184 registers, 16 over the step)
template <int NB, bool FULL, bool CAUSAL, bool SPLIT, bool TAIL = false, bool DROP = false>
__global__ __launch_bounds__(64 * WAVES16, (TAIL && NB == 2 && !DROP) ? 3 : 1) void attn16_kernel(const Geo16 geo) {
  static_assert(FULL || !SPLIT, "DIAG items are independent blocks already");
  static_assert(!TAIL || (FULL && CAUSAL && !SPLIT), "TAIL: one causal sequence of 16 NB + 1 tokens per item");
  constexpr int VROWS = NB * 16;
  __shared__ __attribute__((aligned(16))) float smem_all[WAVES16 * (VROWS * DH16 + VROWS)];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* Vs = smem_all + wave * (VROWS * DH16 + VROWS);
  int* kmeta = reinterpret_cast<int*>(Vs + VROWS * DH16);
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4;
  const int H = geo.H, L = geo.L, d = H * DH16;
  const int64_t ld = 3 * (int64_t)d;
  const int stride = gridDim.x * WAVES16;
  const int n_units = SPLIT ? geo.n_items * NB : geo.n_items;
  int unit = blockIdx.x * WAVES16 + wave;
  if (unit >= n_units) return;

  // does query block qb use key block kb
  auto used = [&](int kb, int qb) __attribute__((always_inline)) { return FULL ? (!CAUSAL || kb <= qb) : kb == qb; };
  // token of block b, local row r of an item that starts at token t0; -1 = no such row
  auto row_token = [&](int t0, int b, int r) __attribute__((always_inline)) {
    const int local = FULL ? b * 16 + r : r;
    const int limit = FULL ? L : geo.P * L;
    const int tok = FULL ? t0 + local : t0 + b * geo.P * L + r;
    return (local < limit && tok < geo.n_tokens) ? tok : -1;
  };
  // a valid token to read in place of an absent row of block b (such rows are masked / never stored)
  auto spare_token = [&](int t0, int b) __attribute__((always_inline)) {
    int last = FULL ? t0 + L - 1 : t0 + b * geo.P * L + geo.P * L - 1;
    if (last > geo.n_tokens - 1) last = geo.n_tokens - 1;
    return last < 0 ? 0 : last;
  };
  // unit -> first token of its item, head, and (SPLIT) its query block: the long blocks of a causal item go first
  auto unit_geo = [&](int u, int& t0, int& head, int& qb0) __attribute__((always_inline)) {
    unsigned it = (unsigned)u;
    qb0 = 0;
    if (SPLIT) {
      it = (unsigned)u / (unsigned)NB;
      qb0 = NB - 1 - (int)((unsigned)u - it * (unsigned)NB);
    }
    if (geo.reverse) it = (unsigned)geo.n_items - 1u - it;
    const unsigned grp = it / (unsigned)H;
    head = (int)(it - grp * (unsigned)H);
    t0 = (int)(grp * (unsigned)geo.rows_per_item);
  };
  // row metadata, the same encoding for keys and queries: -1 = absent, else (sequence in block << 8) | position
  auto row_meta = [&](int b) __attribute__((always_inline)) {
    const int local = FULL ? b * 16 + li : li;
    return ((FULL ? 0 : local / L) << 8) | (FULL ? local : local % L);
  };

  f32x4 kf[NB][4], qc[4];
  int pad[NB];  // key metadata of this lane's row in every block (-1 also for padded keys)
  auto load_k = [&](int u) __attribute__((always_inline)) {
    int t0, head, qb0;
    unit_geo(u, t0, head, qb0);
    const float* base = geo.qkv + head * DH16 + 4 * lg + d;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (SPLIT && !used(b, qb0)) continue;
      const int tok = row_token(t0, b, li);
      const int tk = tok >= 0 ? tok : spare_token(t0, b);
      const float* row = base + (int64_t)tk * ld;
#pragma unroll
      for (int c = 0; c < 4; ++c) kf[b][c] = *reinterpret_cast<const f32x4*>(row + 16 * c);
      pad[b] = (tok >= 0 && geo.kpm[tk] == 0) ? row_meta(b) : -1;
    }
  };
  auto load_q = [&](int u, int b) __attribute__((always_inline)) {
    int t0, head, qb0;
    unit_geo(u, t0, head, qb0);
    const int tok = row_token(t0, b, li);  // (TAIL, b == NB: the tail row for li == 0, and spare_token = the same row for the other lanes)
    const float* row = geo.qkv + head * DH16 + 4 * lg + (int64_t)(tok >= 0 ? tok : spare_token(t0, b)) * ld;
#pragma unroll
    for (int c = 0; c < 4; ++c) qc[c] = *reinterpret_cast<const f32x4*>(row + 16 * c);
  };
  // TAIL: the last token's query, key and value rows, ONE channel per lane (3 registers; in the MFMA layout they would be 16 each and cost
  // the kernel a wave per SIMD): the self score is a wave reduction, the value row reaches the output layout through 16 lane shuffles
  float q1 = 0.f, k1 = 0.f, v1 = 0.f;
  int tail_pad = 1;
  auto load_tail = [&](int u) __attribute__((always_inline)) {
    int t0, head, qb0;
    unit_geo(u, t0, head, qb0);
    const int tk = t0 + L - 1;  // items are whole sequences of the batch: always a valid token
    const float* row = geo.qkv + (int64_t)tk * ld + head * DH16 + lane;
    q1 = row[0]; k1 = row[d]; v1 = row[2 * d];
    tail_pad = geo.kpm[tk];
  };
  auto load_v = [&](int u) __attribute__((always_inline)) {
    int t0, head, qb0;
    unit_geo(u, t0, head, qb0);
    // V rows of the item: 4 rows (1 KB) per LDS-DMA instruction, chunk slot q ^ (row & 15)
#pragma unroll
    for (int i = 0; i < NB * 4; ++i) {
      if (SPLIT && !used(i >> 2, qb0)) continue;
      const int row = 4 * i + (lane >> 4), slot = lane & 15;
      int tok = row_token(t0, row >> 4, row & 15);
      if (tok < 0) tok = t0 < geo.n_tokens ? t0 : geo.n_tokens - 1;
      const float* g = geo.qkv + (int64_t)tok * ld + 2 * d + head * DH16 + (slot ^ (row & 15)) * 4;
      __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(Vs + i * 256), 16, 0, 0);
    }
  };
  {
    int t0, head, qb0;
    unit_geo(unit, t0, head, qb0);
    load_k(unit);
    load_q(unit, qb0);
    load_v(unit);
    if (TAIL) load_tail(unit);
  }
  for (;;) {
    int t0, head, qb0;
    unit_geo(unit, t0, head, qb0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this unit's fragments and V tile have landed (wave-local)
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (lane < 16 && (!SPLIT || used(b, qb0))) kmeta[b * 16 + lane] = pad[b];
    const int n_unit = unit + stride;
    const bool have_next = n_unit < n_units;

    // one query block: qb is a constant of the unrolled loop below, or the unit's block under SPLIT
    auto pass = [&](const int qb, const bool last) __attribute__((always_inline)) {
      // ---- S^T blocks: st[kb][r] = score of key kb*16 + 4*lg + r against query qb*16 + li
      f32x4 st[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (used(kb, qb)) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) st[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kb][c][e], qc[c][e], st[kb], 0, 0, 0);
        }
      }
      // TAIL pass (qb == NB): the tail query against its own key (one channel per lane, summed over the wave)
      const bool tail_pass = TAIL && qb == NB;
      float s_tail = 0.f;
      if (tail_pass) s_tail = tail_pad == 0 ? wave_sum(q1 * k1) * geo.scale : -1e30f;
      // the Q registers are dead (after the last block the K registers too): next loads go under the softmax / P·V
      if (!last) {
        load_q(unit, qb + 1);
      } else if (have_next) {
        int nt0, nhead, nqb0;
        unit_geo(n_unit, nt0, nhead, nqb0);
        load_k(n_unit);
        load_q(n_unit, nqb0);
      }
      // ---- mask + softmax (a query's scores: 4 per key block, over the 4 lanes lg = 0..3)
      const int tok = row_token(t0, qb, li);
      const int mq = tok >= 0 ? row_meta(qb) : -1;
      const int q_seq = mq >> 8, q_pos = mq & 0xff;
      float m = -1e30f;
      int kpos[DROP ? NB : 1][4];  // DROP: the keys' positions inside their sequences (the mask index)
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        if (!used(kb, qb)) continue;
        const int4 km = *reinterpret_cast<const int4*>(kmeta + kb * 16 + 4 * lg);
        const int kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (kmv[r] >= 0) & ((kmv[r] >> 8) == q_seq) & (!CAUSAL || (kmv[r] & 0xff) <= q_pos);
          if (DROP) kpos[kb][r] = kmv[r] & 0xff;
          st[kb][r] = ok ? st[kb][r] * geo.scale : -1e30f;
          m = fmaxf(m, st[kb][r]);
        }
      }
      m = groups_max(m);
      if (tail_pass) m = fmaxf(m, s_tail);
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        if (!used(kb, qb)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[kb][r] > -1e29f ? __expf(st[kb][r] - m) : 0.f;
          st[kb][r] = p;
          sum += p;
        }
      }
      sum = groups_sum(sum);
      float p_tail = 0.f;
      if (tail_pass) {
        p_tail = s_tail > -1e29f ? __expf(s_tail - m) : 0.f;
        sum += p_tail;
      }
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;  // fully masked row -> zeros
      if (DROP) {  // train-mode dropout of the probabilities: the denominator keeps the undropped sum (as attn.hip)
        const uint64_t key = stlt_drop_key(geo.dr, geo.site);
        const uint64_t qidx = (((uint64_t)(tok >= 0 ? tok : 0)) * (uint64_t)H + (uint64_t)head) << 8;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
          if (!used(kb, qb)) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) st[kb][r] = stlt_keep_k(geo.dr.thr, key, qidx | (uint64_t)kpos[kb][r]) ? st[kb][r] * geo.dr.scale : 0.f;
        }
        if (tail_pass) p_tail = stlt_keep_k(geo.dr.thr, key, qidx | (uint64_t)(L - 1)) ? p_tail * geo.dr.scale : 0.f;
      }
      // ---- O^T[channel][query] += V^T·P^T: MFMA step (kb, r) sums keys kb*16 + 4g + r over g
      f32x4 o[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) o[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        if (!used(kb, qb)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kb * 16 + 4 * lg + r;  // key row inside the item's V tile
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            const float v = Vs[j * DH16 + (((cb * 4 + (li >> 2)) ^ (j & 15)) * 4) + (li & 3)];
            o[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, st[kb][r], o[cb], 0, 0, 0);
          }
        }
      }
      if (tail_pass) {  // the tail's own value row: channel 16 cb + 4 lg + r sits in lane 16 cb + 4 lg + r
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[cb][r] = fmaf(p_tail, __shfl(v1, cb * 16 + 4 * lg + r, 64), o[cb][r]);
      }
      if (last && have_next) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the unit's last V reads have returned: the tile can be refilled
        load_v(n_unit);
        if (TAIL) load_tail(n_unit);
      }
      // ---- store: lane (query li, lg) holds channels cb*16 + 4*lg .. +3 of its query
      if (tok >= 0) {
        float* dst = geo.ctx + (int64_t)tok * d + head * DH16 + 4 * lg;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<f32x4*>(dst + 16 * cb) = o[cb] * inv;
      }
    };
    if (SPLIT) {
      pass(qb0, true);
    } else {
#pragma unroll
      for (int qb = 0; qb < (TAIL ? NB + 1 : NB); ++qb) pass(qb, qb == (TAIL ? NB : NB - 1));
    }
    if (!have_next) break;
    unit = n_unit;
  }
}

// wave slots of the device for one instantiation (workgroups per CU x CUs x waves per workgroup)
template <int NB, bool FULL, bool CAUSAL, bool SPLIT, bool TAIL = false, bool DROP = false>
int64_t wg_capacity16() {
  static StltPerDeviceInt occ;
  int& wg_per_cu = occ.ref();
  if (wg_per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, attn16_kernel<NB, FULL, CAUSAL, SPLIT, TAIL, DROP>, 64 * WAVES16, 0) != hipSuccess || wg_per_cu <= 0)
      wg_per_cu = 1;
  }
  return (int64_t)wg_per_cu * stlt_device_cus();
}

template <int NB, bool FULL, bool CAUSAL, bool SPLIT, bool TAIL, bool DROP>
int launch16_drop(const Geo16& g, hipStream_t s) {
  const int64_t n_units = SPLIT ? (int64_t)g.n_items * NB : g.n_items;
  int64_t n_wg = (n_units + WAVES16 - 1) / WAVES16;
  const int64_t cap = wg_capacity16<NB, FULL, CAUSAL, SPLIT, TAIL, DROP>();
  if (n_wg > cap) n_wg = cap;
  hipLaunchKernelGGL((attn16_kernel<NB, FULL, CAUSAL, SPLIT, TAIL, DROP>), dim3((unsigned)n_wg), dim3(64 * WAVES16), 0, s, g);
  return stlt_check_launch("attn16_kernel");
}
template <int NB, bool FULL, bool CAUSAL, bool SPLIT, bool TAIL = false>
int launch16_as(const Geo16& g, hipStream_t s) {
  return g.dr.thr ? launch16_drop<NB, FULL, CAUSAL, SPLIT, TAIL, true>(g, s) : launch16_drop<NB, FULL, CAUSAL, SPLIT, TAIL, false>(g, s);
}

// FULL launches with fewer items than `split_below` x the device's wave slots are cut into (item, query block) units
template <int NB, bool CAUSAL>
int launch16_full(const Geo16& g, hipStream_t s) {
  static const double split_below = [] { const char* e = getenv("STLT_ATTN16_SPLIT_BELOW"); return e ? atof(e) : 0.5; }();
  const int64_t slots = (g.dr.thr ? wg_capacity16<NB, true, CAUSAL, false, false, true>() : wg_capacity16<NB, true, CAUSAL, false>()) * WAVES16;
  if ((double)g.n_items < split_below * (double)slots && (int64_t)g.n_items * NB <= 0x7fffffffLL) return launch16_as<NB, true, CAUSAL, true>(g, s);
  return launch16_as<NB, true, CAUSAL, false>(g, s);
}

}  // namespace

// *taken = true when the launch was made (return value: 0 or the error), false when the shape is not this kernel's (the
// caller then uses attn.hip).
int launch_attn16(const float* qkv, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, float* ctx, int reverse, hipStream_t s,
                  bool* taken, StltDrop dr, uint32_t site) {
  *taken = false;
  static const int enabled = [] { const char* e = getenv("STLT_ATTN16"); return e ? atoi(e) : 1; }();
  static const int drop_on = [] { const char* e = getenv("STLT_ATTN16_DROPOUT"); return e ? atoi(e) : 1; }();
  if (!enabled || L < 1 || L > 64 || (dr.thr && !drop_on)) return 0;
  const int64_t n_tokens = S * L;
  if (n_tokens > 0x7fffffffLL || H > 65535) return 0;
  Geo16 g;
  g.qkv = qkv; g.kpm = kpm; g.ctx = ctx;
  g.n_tokens = (int)n_tokens; g.L = (int)L; g.H = (int)H;
  g.reverse = reverse;
  g.scale = 0.125f;  // 1 / sqrt(64)
  g.dr = dr; g.site = site;
  int rc;
  if (L <= 16) {
    if (causal) return 0;  // short causal sequences: not a shape of the path (the temporal pass has T frames), keep attn.hip
    g.P = (int)(16 / L);
    constexpr int NB = 3;
    g.rows_per_item = NB * g.P * (int)L;
    const int64_t items = ((n_tokens + g.rows_per_item - 1) / g.rows_per_item) * H;
    if (items > 0x7fffffffLL) return 0;
    g.n_items = (int)items;
    rc = launch16_as<NB, false, false, false>(g, s);
  } else {
    g.P = 1;
    g.rows_per_item = (int)L;
    const int64_t items = S * H;
    if (items > 0x7fffffffLL) return 0;
    g.n_items = (int)items;
    // causal sequences of 16 NB + 1 tokens (the reference's T = 17 / 33; 49): NB full blocks + the tail row (STLT_ATTN16_TAIL=0: as NB + 1 blocks)
    static const int tail_on = [] { const char* e = getenv("STLT_ATTN16_TAIL"); return e ? atoi(e) : 1; }();
    if (causal && tail_on && L % 16 == 1) {
      rc = L == 17 ? launch16_as<1, true, true, false, true>(g, s) : (L == 33 ? launch16_as<2, true, true, false, true>(g, s) : launch16_as<3, true, true, false, true>(g, s));
      *taken = true;
      return rc;
    }
    if (L <= 32) rc = causal ? launch16_full<2, true>(g, s) : launch16_full<2, false>(g, s);
    else if (L <= 48) rc = causal ? launch16_full<3, true>(g, s) : launch16_full<3, false>(g, s);
    else rc = causal ? launch16_full<4, true>(g, s) : launch16_full<4, false>(g, s);
  }
  *taken = true;
  return rc;
}
