// K3, 16-row tiles: the attention core for self-attention over short sequences (L <= 48 tokens on a packed QKV buffer,
// no dropout) on v_mfma_f32_16x16x4_f32.  attn.hip's 32x32 tiles are exact for L = 32 / 64 and for 4 x 7-token frames, but
// a 36-token sequence (cfg4: N = 36) costs them four (query tile, key tile) steps of which 68 % is padding, and a wave
// needs 64 dependent 64-cycle MFMAs per (sequence, head).  Here an item is up to three 16-row blocks:
//   FULL  (16 < L <= 48): one sequence = NB = ceil(L/16) blocks; every (query block, key block) pair, or the lower
//         triangle when causal — cfg2 temporal (T = 32): 3 pairs instead of a 32x32 tile; cfg4 spatial (N = 36): 9 pairs of
//         16x16 instead of 4 of 32x32;
//   DIAG  (L <= 16): P = floor(16/L) whole sequences per block, NB independent blocks per item, diagonal pairs only —
//         cfg2 spatial (N = 7): 2 frames per block.
// Same dataflow as attn.hip: S^T = K·Q^T with swapped operands (a query's scores sit in 4 lanes x 4 registers per key
// block: row max / sum are in-register + two shuffles, and the probabilities are already the B operand of O^T += V^T·P^T),
// Q and K fragments straight from global memory in operand shape, V by LDS-DMA into a swizzled tile, the next item's
// loads issued as soon as Q·K^T has consumed the fragment registers.  All of an item's scores fit in registers (<= 36),
// so the softmax is one pass (no running maximum).  The output leaves without an LDS transpose: lane (query, g) holds 4
// consecutive channels per channel block = one 16-byte store.  Bit-for-bit it is a different summation order than
// attn.hip (16-key blocks); both are tested against the same fp64 oracle.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int DH16 = 64;
// independent waves per workgroup: 4, or 2 with three blocks per item (24.8 KB of LDS per wave: 2 waves x 3 workgroups fit a CU)
template <int NB> struct Waves16 { static constexpr int value = NB == 3 ? 2 : 4; };

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
};

template <int NB, bool FULL, bool CAUSAL>
__global__ __launch_bounds__(64 * Waves16<NB>::value) void attn16_kernel(const Geo16 geo) {
  constexpr int WAVES16 = Waves16<NB>::value;
  constexpr int VROWS = NB * 16;
  __shared__ __attribute__((aligned(16))) float smem_all[WAVES16 * (2 * VROWS * DH16 + VROWS)];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* smem = smem_all + wave * (2 * VROWS * DH16 + VROWS);
  int* kmeta = reinterpret_cast<int*>(smem + 2 * VROWS * DH16);
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4;
  const int H = geo.H, L = geo.L, d = H * DH16;
  const int64_t ld = 3 * (int64_t)d;
  const int stride = gridDim.x * WAVES16;
  int item = blockIdx.x * WAVES16 + wave;
  if (item >= geo.n_items) return;

  // token of block b, local row r of an item that starts at token t0; -1 = no such row
  auto row_token = [&](int t0, int b, int r) {
    const int local = FULL ? b * 16 + r : r;
    const int limit = FULL ? L : geo.P * L;
    const int tok = FULL ? t0 + local : t0 + b * geo.P * L + r;
    return (local < limit && tok < geo.n_tokens) ? tok : -1;
  };
  auto item_geo = [&](int it, int& t0, int& head) {
    const unsigned u = geo.reverse ? (unsigned)geo.n_items - 1u - (unsigned)it : (unsigned)it;
    const unsigned grp = u / (unsigned)H;
    head = (int)(u - grp * (unsigned)H);
    t0 = (int)(grp * (unsigned)geo.rows_per_item);
  };

  f32x4 qf[NB][4], kf[NB][4];
  int pad[NB];  // key metadata of this lane's row in every block: -1 = masked / absent, else (sequence << 8) | position
  auto load_item = [&](int it, int buf) {
    int t0, head;
    item_geo(it, t0, head);
    const float* base = geo.qkv + head * DH16 + 4 * lg;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int tok = row_token(t0, b, li);
      int last = FULL ? t0 + L - 1 : t0 + b * geo.P * L + geo.P * L - 1;
      if (last > geo.n_tokens - 1) last = geo.n_tokens - 1;
      if (last < 0) last = 0;
      const int tk = tok >= 0 ? tok : last;  // rows past the end re-read a valid row; they are masked / never stored
      const float* row = base + (int64_t)tk * ld;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qf[b][c] = *reinterpret_cast<const f32x4*>(row + 16 * c);
        kf[b][c] = *reinterpret_cast<const f32x4*>(row + d + 16 * c);
      }
      const int local = FULL ? b * 16 + li : li;
      pad[b] = (tok >= 0 && geo.kpm[tk] == 0) ? (((FULL ? 0 : local / L) << 8) | (FULL ? local : local % L)) : -1;
    }
    // V rows of the item: 4 rows (1 KB) per LDS-DMA instruction, chunk slot q ^ (row & 15)
    float* Vs = smem + buf * VROWS * DH16;
#pragma unroll
    for (int i = 0; i < NB * 4; ++i) {
      const int row = 4 * i + (lane >> 4), slot = lane & 15;
      int tok = row_token(t0, row >> 4, row & 15);
      if (tok < 0) tok = t0 < geo.n_tokens ? t0 : geo.n_tokens - 1;
      const float* g = geo.qkv + (int64_t)tok * ld + 2 * d + head * DH16 + (slot ^ (row & 15)) * 4;
      __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(Vs + i * 256), 16, 0, 0);
    }
  };

  int buf = 0;
  load_item(item, 0);
  for (;;) {
    int t0, head;
    item_geo(item, t0, head);
    float* Vs = smem + buf * VROWS * DH16;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this item's fragments and V tile have landed (wave-local)
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (lane < 16) kmeta[b * 16 + lane] = pad[b];

    // ---- S^T blocks: st[qb][kb][r] = score of key kb*16 + 4*lg + r against query qb*16 + li
    f32x4 st[NB][NB];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        const bool use = FULL ? (!CAUSAL || kb <= qb) : kb == qb;
        st[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (use) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kb][c][e], qf[qb][c][e], st[qb][kb], 0, 0, 0);
        }
      }
    // the fragment registers are dead: put the next item's loads in flight under the softmax / P·V / stores
    const int n_item = item + stride;
    const bool have_next = n_item < geo.n_items;
    int my_q[NB];  // this lane's query metadata per block (same encoding as the keys'), taken before the registers are reloaded
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int local = FULL ? b * 16 + li : li;
      my_q[b] = row_token(t0, b, li) >= 0 ? (((FULL ? 0 : local / L) << 8) | (FULL ? local : local % L)) : -1;
    }
    if (have_next) load_item(n_item, buf ^ 1);

    // ---- mask + softmax (all of a query's scores are in registers: 4 per key block, over the 4 lanes lg = 0..3)
    float inv[NB];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
      const int q_seq = my_q[qb] >> 8, q_pos = my_q[qb] & 0xff;
      float m = -1e30f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        const bool use = FULL ? (!CAUSAL || kb <= qb) : kb == qb;
        if (!use) continue;
        const int4 km = *reinterpret_cast<const int4*>(kmeta + kb * 16 + 4 * lg);
        const int kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (kmv[r] >= 0) & ((kmv[r] >> 8) == q_seq) & (!CAUSAL || (kmv[r] & 0xff) <= q_pos);
          st[qb][kb][r] = ok ? st[qb][kb][r] * geo.scale : -1e30f;
          m = fmaxf(m, st[qb][kb][r]);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        const bool use = FULL ? (!CAUSAL || kb <= qb) : kb == qb;
        if (!use) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[qb][kb][r] > -1e29f ? __expf(st[qb][kb][r] - m) : 0.f;
          st[qb][kb][r] = p;
          sum += p;
        }
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      inv[qb] = sum > 0.f ? 1.0f / sum : 0.f;  // fully masked row -> zeros
    }

    // ---- O^T[channel][query] += V^T·P^T: MFMA step (kb, r) sums keys kb*16 + 4g + r over g; one V value per lane and
    // step feeds every query block that uses the key block
    f32x4 o[NB][4];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) o[qb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = kb * 16 + 4 * lg + r;  // key row inside the item's V tile
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          const float v = Vs[j * DH16 + (((cb * 4 + (li >> 2)) ^ (j & 15)) * 4) + (li & 3)];
#pragma unroll
          for (int qb = 0; qb < NB; ++qb) {
            const bool use = FULL ? (!CAUSAL || kb <= qb) : kb == qb;
            if (use) o[qb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, st[qb][kb][r], o[qb][cb], 0, 0, 0);
          }
        }
      }

    // ---- stores: lane (query li, lg) holds channels cb*16 + 4*lg .. +3 of its query
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
      const int tok = row_token(t0, qb, li);
      if (tok >= 0) {
        float* dst = geo.ctx + (int64_t)tok * d + head * DH16 + 4 * lg;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<f32x4*>(dst + 16 * cb) = o[qb][cb] * inv[qb];
      }
    }
    if (!have_next) break;
    item = n_item;
    buf ^= 1;
  }
}

template <int NB, bool FULL, bool CAUSAL>
int launch16(const Geo16& g, hipStream_t s) {
  constexpr int WAVES16 = Waves16<NB>::value;
  static StltPerDeviceInt occ;
  int& wg_per_cu = occ.ref();
  if (wg_per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, attn16_kernel<NB, FULL, CAUSAL>, 64 * WAVES16, 0) != hipSuccess || wg_per_cu <= 0)
      wg_per_cu = 1;
  }
  int64_t n_wg = ((int64_t)g.n_items + WAVES16 - 1) / WAVES16;
  const int64_t cap = (int64_t)wg_per_cu * stlt_device_cus();
  if (n_wg > cap) n_wg = cap;
  hipLaunchKernelGGL((attn16_kernel<NB, FULL, CAUSAL>), dim3((unsigned)n_wg), dim3(64 * WAVES16), 0, s, g);
  return stlt_check_launch("attn16_kernel");
}

}  // namespace

// Returns 1 when the launch was taken, 0 when the shape is not this kernel's (the caller then uses attn.hip), < 0 / hip error on failure.
int launch_attn16(const float* qkv, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, float* ctx, int reverse, hipStream_t s) {
  static const int enabled = [] { const char* e = getenv("STLT_ATTN16"); return e ? atoi(e) : 1; }();
  if (!enabled || L < 1 || L > 48) return 0;
  const int64_t n_tokens = S * L;
  if (n_tokens > 0x7fffffffLL || H > 65535) return 0;
  Geo16 g;
  g.qkv = qkv; g.kpm = kpm; g.ctx = ctx;
  g.n_tokens = (int)n_tokens; g.L = (int)L; g.H = (int)H;
  g.reverse = reverse;
  g.scale = 0.125f;  // 1 / sqrt(64)
  int rc;
  if (L <= 16) {
    if (causal) return 0;  // short causal sequences: not a shape of the path (the temporal pass has T frames), keep attn.hip
    g.P = (int)(16 / L);
    constexpr int NB = 3;
    g.rows_per_item = NB * g.P * (int)L;
    const int64_t items = ((n_tokens + g.rows_per_item - 1) / g.rows_per_item) * H;
    if (items > 0x7fffffffLL) return 0;
    g.n_items = (int)items;
    rc = launch16<NB, false, false>(g, s);
  } else {
    g.P = 1;
    g.rows_per_item = (int)L;
    const int64_t items = S * H;
    if (items > 0x7fffffffLL) return 0;
    g.n_items = (int)items;
    if (L <= 32) rc = causal ? launch16<2, true, true>(g, s) : launch16<2, true, false>(g, s);
    else rc = causal ? launch16<3, true, true>(g, s) : launch16<3, true, false>(g, s);
  }
  return rc == 0 ? 1 : rc;
}
