// Attention core backward on v_mfma_f32_16x16x4_f32 for the groups the training step is made of: sequences of at most 64
// tokens on the packed QKV buffer (cfg2: T = 32 frames per clip, N = 7 objects per frame; cfg4: T = 64, N = 36; the fusion
// models' 33 appearance tokens), with the key-padding / causal masks, the dropout on the attention probabilities and the
// in-projection bias gradient of backward.hip's attn_bwd_kernel, which stays the kernel of the ragged (skip-padding) layout.
//   P = softmax(scale·Q·Kᵀ + mask), Pd = P∘D (dropout), O = Pd·V
//   dV = Pdᵀ·dO ; dPd = dO·Vᵀ ; dP = dPd∘D ; dS = P∘(dP − rowsum(P∘dP)) ; dQ = scale·dS·K ; dK = scale·dSᵀ·Q
// One wave per (item, head); an item is NB 16-row blocks:
//   FULL (16 < L <= 16 NB, NB = 2 / 3 / 4): one sequence, every (query block, key block) pair — the lower triangle when causal;
//   DIAG (L <= 16, NB = 2): floor(16/L) whole sequences per block, diagonal pairs only (2 frames per block at N = 7).
// Q, K, V, dO tiles (16 NB x 64 each) arrive by LDS-DMA in the swizzled layout of attn16.hip; 4 / 3 / 2 waves per workgroup at
// NB = 2 / 3 / 4 (32 / 48 / 64 KB of LDS per wave), one workgroup per CU.  Scores and dPd are computed
// transposed — Sᵀ = K·Qᵀ, dPdᵀ = V·dOᵀ — so that a query's row sits in 4 lanes x 4 registers per key block: the softmax
// and the rowsum are in-register + two shuffles, and dSᵀ is already the B operand of dQᵀ = Kᵀ·dSᵀ.  dK and dV contract
// over the queries and want dS / Pd with the keys in the lane index instead: both go through an LDS transpose (dS in the V
// tile's space, Pd in the K tile's: both are dead by then).  Outputs leave as 16-byte stores (a lane holds 4 consecutive channels).
// A wave keeps one head for its whole life (wave w: head w % H, items w / H, w / H + chunks, ...), so the column sums of
// dQ / dK / dV (the in-projection bias gradient) accumulate in registers per lane and are reduced once at the end into
// slab w / H of the scratch buffer, which launch_reduce_slabs adds in fixed order: bitwise reproducible.
#include <cstdlib>
#include "common.h"
#include "wave_dpp.h"

namespace {

constexpr int BD = 64;      // head dim
constexpr int waves_for(int nb) { return nb == 2 ? 4 : nb == 3 ? 3 : 2; }  // independent waves per workgroup (16 NB KB of LDS each)

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

struct BwdGeo {
  const float* qkv;    // packed (tokens, 3*H*64): q | k | v
  const float* dctx;   // (tokens, H*64)
  const uint8_t* kpm;  // one byte per token: 1 = padded key
  float* dqkv;         // (tokens, 3*H*64)
  float* cs;           // nullable: column-sum slabs [chunks][3*H*64]
  int n_tokens, L, H, P;
  int rows_per_item;   // FULL: L; DIAG: 2 * P * L
  int n_items, chunks;
  float scale;
  StltDrop dr;
  uint32_t site;
  // RAGGED (skip-padding layout): item g = rows [grp_ptr[g], grp_ptr[g+1]) of the compacted buffer — whole segments, at most 16 NB
  // rows; a row's segment starts at seg_start[row].  No key-padding bytes: every row is real.
  const int* grp_ptr;
  const int* seg_start;
};

template <int NB, bool FULL, bool CAUSAL, bool DROP, bool RAGGED = false>
__global__ __launch_bounds__(64 * waves_for(NB)) void attn_bwd16_kernel(const BwdGeo geo) {
  static_assert(!RAGGED || FULL, "ragged items are walked as FULL items: segments may straddle the 16-row blocks");
  constexpr int BROWS = 16 * NB, TILE = BROWS * BD, BWAVES = waves_for(NB);
  __shared__ __attribute__((aligned(16))) float smem_all[BWAVES * (4 * TILE + BROWS)];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* Qs = smem_all + wave * (4 * TILE + BROWS);
  float* Ks = Qs + TILE;
  float* Vs = Ks + TILE;
  float* Gs = Vs + TILE;
  int* kmeta = reinterpret_cast<int*>(Gs + TILE);
  float* Tds = Vs;             // dS[key][query], BROWS x BROWS, once the V tile is dead
  float* Tpd = Ks;             // Pd[key][query], once the K tile is dead (after the last dQ)
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4;
  const int H = geo.H, L = geo.L, d = H * BD;
  const int64_t ld = 3 * (int64_t)d;
  const int w = blockIdx.x * BWAVES + wave;
  const int head = w % H, chunk = w / H;
  if (chunk >= geo.chunks) return;  // the grid is rounded up to whole workgroups

  auto used = [&](int kb, int qb) __attribute__((always_inline)) { return FULL ? (!CAUSAL || kb <= qb) : kb == qb; };
  int item_rows = L;  // RAGGED: rows of the current item
  // token of block b, local row r of the item that starts at token t0; -1 = no such row
  auto row_token = [&](int t0, int b, int r) __attribute__((always_inline)) {
    const int local = FULL ? b * 16 + r : r;
    const int limit = RAGGED ? item_rows : FULL ? L : geo.P * L;
    const int tok = FULL ? t0 + local : t0 + b * geo.P * L + r;
    return (local < limit && tok < geo.n_tokens) ? tok : -1;
  };
  // (sequence inside the item << 8) | position inside the sequence, of a real row (token tok, local index `local`)
  auto meta_of = [&](int t0, int tok, int local) __attribute__((always_inline)) {
    if (RAGGED) { const int ss = geo.seg_start[tok]; return ((ss - t0) << 8) | (tok - ss); }
    return ((FULL ? 0 : local / L) << 8) | (FULL ? local : local % L);
  };
  // float offset of channel chunk c (4 floats) of row `row` in a swizzled tile
  auto swz = [&](int row, int c) __attribute__((always_inline)) { return row * BD + ((c ^ (row & 15)) << 2); };

  float csum[3][4][4];  // per-lane partial column sums of dQ / dK / dV: [matrix][channel block][register]
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) csum[a][b][c] = 0.f;

  for (int item = chunk; item < geo.n_items; item += geo.chunks) {
    const int t0 = RAGGED ? geo.grp_ptr[item] : item * geo.rows_per_item;
    if (RAGGED) item_rows = geo.grp_ptr[item + 1] - t0;
    // ---- tiles by LDS-DMA: 4 rows (1 KB) per instruction, chunk slot q holds channels 4*(q ^ (row & 15))..
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous item's LDS reads have returned
#pragma unroll
    for (int i = 0; i < BROWS / 4; ++i) {
      const int row = 4 * i + (lane >> 4), slot = lane & 15;
      int tok = row_token(t0, row >> 4, row & 15);
      if (tok < 0) tok = t0 < geo.n_tokens ? t0 : geo.n_tokens - 1;  // absent rows re-read a valid row; they are masked / never stored
      const int ch = head * BD + (slot ^ (row & 15)) * 4;
      const float* gq = geo.qkv + (int64_t)tok * ld + ch;
      __builtin_amdgcn_global_load_lds((glb_void_ptr)gq, (lds_void_ptr)(Qs + i * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_ptr)(gq + d), (lds_void_ptr)(Ks + i * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_ptr)(gq + 2 * d), (lds_void_ptr)(Vs + i * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_ptr)(geo.dctx + (int64_t)tok * d + ch), (lds_void_ptr)(Gs + i * 256), 16, 0, 0);
    }
    if (lane < BROWS) {
      const int b = lane >> 4, r = lane & 15;
      const int tok = row_token(t0, b, r);
      const int local = FULL ? b * 16 + r : r;
      kmeta[lane] = (tok >= 0 && (RAGGED || geo.kpm[tok] == 0)) ? meta_of(t0, tok, local) : -1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // wave-local: tiles and key metadata are in LDS

    f32x4 ds[NB][NB], pd[NB][NB];  // dSᵀ and Pdᵀ of (query block, key block): lane = query li, register r = key 4*lg + r
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
      // ---- Sᵀ = K·Qᵀ and dPdᵀ = V·dOᵀ for this query block
      const int qrow = qb * 16 + li;
      f32x4 qf[4], gf[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qf[c] = *reinterpret_cast<const f32x4*>(Qs + swz(qrow, 4 * c + lg));
        gf[c] = *reinterpret_cast<const f32x4*>(Gs + swz(qrow, 4 * c + lg));
      }
      f32x4 st[NB], dp[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!used(kb, qb)) continue;
        const int krow = kb * 16 + li;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + swz(krow, 4 * c + lg));
          const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + swz(krow, 4 * c + lg));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            st[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[c][e], st[kb], 0, 0, 0);
            dp[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], gf[c][e], dp[kb], 0, 0, 0);
          }
        }
      }
      // ---- mask, softmax, dropout, dS
      const int qtok = row_token(t0, qb, li);
      const int mq = qtok >= 0 ? meta_of(t0, qtok, FULL ? qb * 16 + li : li) : -1;
      const int q_seq = mq >> 8, q_pos = mq & 0xff;
      int kpos[NB][4];
      float m = -1e30f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        if (!used(kb, qb)) continue;
        const int4 km = *reinterpret_cast<const int4*>(kmeta + kb * 16 + 4 * lg);
        const int kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (mq >= 0) & (kmv[r] >= 0) & ((kmv[r] >> 8) == q_seq) & (!CAUSAL || (kmv[r] & 0xff) <= q_pos);
          kpos[kb][r] = kmv[r] & 0xff;
          st[kb][r] = ok ? st[kb][r] * geo.scale : -1e30f;
          m = fmaxf(m, st[kb][r]);
        }
      }
      m = groups_max(m);
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
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;  // fully masked row -> zeros
      float dsum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        pd[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!used(kb, qb)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[kb][r] * inv;
          float g = dp[kb][r];  // dPd
          float pdv = p;
          if (DROP) {  // the forward multiplied P by the mask before P·V: dP = dPd∘D, and dV sees Pd = P∘D
            const uint64_t idx = ((((uint64_t)(qtok >= 0 ? qtok : 0)) * H + head) << 8) | (uint64_t)kpos[kb][r];
            const bool keep = stlt_keep(geo.dr, geo.site, idx);
            g = keep ? g * geo.dr.scale : 0.f;
            pdv = keep ? p * geo.dr.scale : 0.f;
          }
          st[kb][r] = p;
          dp[kb][r] = g;
          pd[qb][kb][r] = pdv;
          dsum += p * g;
        }
      }
      dsum = groups_sum(dsum);
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        ds[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!used(kb, qb)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[qb][kb][r] = st[kb][r] * (dp[kb][r] - dsum);
      }
      // ---- dQᵀ[channel][query] = Kᵀ·dSᵀ: MFMA step (kb, r) sums keys kb*16 + 4g + r over g
      f32x4 o[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) o[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        if (!used(kb, qb)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kb * 16 + 4 * lg + r;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            const float kv = Ks[j * BD + (((cb * 4 + (li >> 2)) ^ (j & 15)) << 2) + (li & 3)];
            o[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv, ds[qb][kb][r], o[cb], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        o[cb] *= geo.scale;
        if (qtok >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) csum[0][cb][r] += o[cb][r];
        }
      }
      if (qtok >= 0) {
        float* dst = geo.dqkv + (int64_t)qtok * ld + head * BD + 4 * lg;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<f32x4*>(dst + 16 * cb) = o[cb];
      }
    }

    // ---- transposes: dS and Pd with the key in the row, 4 consecutive queries per 16-byte read (V tile is dead)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb * 16 + 4 * lg + r;
          Tds[key * BROWS + qb * 16 + li] = ds[qb][kb][r];
          Tpd[key * BROWS + qb * 16 + li] = pd[qb][kb][r];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- dKᵀ[channel][key] = scale·Qᵀ·dS and dVᵀ[channel][key] = dOᵀ·Pd: MFMA step (qb, r) sums queries qb*16 + 4g + r
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const int ktok = row_token(t0, kb, li);
      f32x4 ok_[4], ov[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) { ok_[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int qb = 0; qb < NB; ++qb) {
        if (!used(kb, qb)) continue;
        const f32x4 bs = *reinterpret_cast<const f32x4*>(Tds + (kb * 16 + li) * BROWS + qb * 16 + 4 * lg);
        const f32x4 bp = *reinterpret_cast<const f32x4*>(Tpd + (kb * 16 + li) * BROWS + qb * 16 + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qb * 16 + 4 * lg + r;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            const int off = q * BD + (((cb * 4 + (li >> 2)) ^ (q & 15)) << 2) + (li & 3);
            ok_[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[off], bs[r], ok_[cb], 0, 0, 0);
            ov[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[off], bp[r], ov[cb], 0, 0, 0);
          }
        }
      }
      if (ktok >= 0) {
        float* dst = geo.dqkv + (int64_t)ktok * ld + d + head * BD + 4 * lg;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          ok_[cb] *= geo.scale;
          *reinterpret_cast<f32x4*>(dst + 16 * cb) = ok_[cb];
          *reinterpret_cast<f32x4*>(dst + d + 16 * cb) = ov[cb];
#pragma unroll
          for (int r = 0; r < 4; ++r) { csum[1][cb][r] += ok_[cb][r]; csum[2][cb][r] += ov[cb][r]; }
        }
      }
    }
  }

  if (geo.cs) {  // lane (li, lg) holds channel cb*16 + 4*lg + r summed over its own rows: add the 16 row lanes, lane li == 0 writes
    float* out = geo.cs + (int64_t)chunk * 3 * d + head * BD;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = csum[a][cb][r];
          x += __shfl_xor(x, 1, 64);
          x += __shfl_xor(x, 2, 64);
          x += __shfl_xor(x, 4, 64);
          x += __shfl_xor(x, 8, 64);
          v[r] = x;
        }
        if (li == 0) *reinterpret_cast<f32x4*>(out + a * d + cb * 16 + 4 * lg) = v;
      }
  }
}

template <int NB, bool FULL, bool CAUSAL>
int launch_bwd16(const BwdGeo& g, int n_wg, hipStream_t s) {
  constexpr int BWAVES = waves_for(NB);
  if (g.dr.thr) hipLaunchKernelGGL((attn_bwd16_kernel<NB, FULL, CAUSAL, true>), dim3((unsigned)n_wg), dim3(64 * BWAVES), 0, s, g);
  else hipLaunchKernelGGL((attn_bwd16_kernel<NB, FULL, CAUSAL, false>), dim3((unsigned)n_wg), dim3(64 * BWAVES), 0, s, g);
  return stlt_check_launch("attn_bwd16_kernel");
}

template <int NB, bool CAUSAL>
int launch_bwd16_ragged(const BwdGeo& g, int n_wg, hipStream_t s) {
  constexpr int BWAVES = waves_for(NB);
  if (g.dr.thr) hipLaunchKernelGGL((attn_bwd16_kernel<NB, true, CAUSAL, true, true>), dim3((unsigned)n_wg), dim3(64 * BWAVES), 0, s, g);
  else hipLaunchKernelGGL((attn_bwd16_kernel<NB, true, CAUSAL, false, true>), dim3((unsigned)n_wg), dim3(64 * BWAVES), 0, s, g);
  return stlt_check_launch("attn_bwd16_kernel(ragged)");
}

}  // namespace

// The same for the ragged (skip-padding) layout: groups of whole segments of at most 64 rows (AttnBwdRagged of common.h).
int launch_attn_bwd16_ragged(const float* qkv, const float* dctx, const AttnBwdRagged& rg, int causal, int64_t H, float* dqkv, StltDrop dr,
                             uint32_t site, float* scratch, int want_colsum, int* chunks_out, hipStream_t s, bool* taken) {
  *taken = false;
  static const int enabled = [] { const char* e = getenv("STLT_ATTN_BWD16"); return e ? atoi(e) : 1; }();
  if (enabled != 1 || rg.max_rows < 1 || rg.max_rows > 64 || rg.n_groups < 1 || rg.n_rows > 0x7fffffffLL || H > 4096) return 0;
  const int nb = rg.max_rows <= 32 ? 2 : rg.max_rows <= 48 ? 3 : 4;
  const int bwaves = waves_for(nb);
  BwdGeo g;
  g.qkv = qkv; g.dctx = dctx; g.kpm = nullptr; g.dqkv = dqkv;
  g.n_tokens = (int)rg.n_rows; g.L = rg.max_rows; g.H = (int)H; g.P = 1; g.rows_per_item = rg.max_rows;
  g.scale = 0.125f;
  g.dr = dr; g.site = site;
  g.grp_ptr = rg.grp_ptr; g.seg_start = rg.seg_start;
  g.n_items = (int)rg.n_groups;
  int64_t chunks = ((int64_t)stlt_device_cus() * bwaves) / H;
  if (chunks < 1) chunks = 1;
  if (chunks > rg.n_groups) chunks = rg.n_groups;
  if (chunks > 256) chunks = 256;
  g.chunks = (int)chunks;
  g.cs = want_colsum ? scratch : nullptr;
  if (chunks_out) *chunks_out = (int)chunks;
  const int n_wg = (int)((chunks * H + bwaves - 1) / bwaves);
  int rc;
  if (nb == 2) rc = causal ? launch_bwd16_ragged<2, true>(g, n_wg, s) : launch_bwd16_ragged<2, false>(g, n_wg, s);
  else if (nb == 3) rc = causal ? launch_bwd16_ragged<3, true>(g, n_wg, s) : launch_bwd16_ragged<3, false>(g, n_wg, s);
  else rc = causal ? launch_bwd16_ragged<4, true>(g, n_wg, s) : launch_bwd16_ragged<4, false>(g, n_wg, s);
  *taken = true;
  return rc;
}

// *taken = true when the launch was made (return value: 0 or the error; column-sum slabs, if asked for, are in `scratch`:
// *chunks_out slabs of 3*H*64 floats, to be added by launch_reduce_slabs), false when the shape is not this kernel's.
int launch_attn_bwd16(const float* qkv, const float* dctx, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, float* dqkv,
                      StltDrop dr, uint32_t site, float* scratch, int want_colsum, int* chunks_out, hipStream_t s, bool* taken) {
  *taken = false;
  static const int enabled = [] { const char* e = getenv("STLT_ATTN_BWD16"); return e ? atoi(e) : 1; }();  // 0: off; 2: sequences of at most 32 tokens only (round-2 coverage, A/B runs)
  if (!enabled || L < 1 || L > 64 || (enabled == 2 && L > 32)) return 0;
  if (L <= 16 && causal) return 0;  // short causal sequences are not a shape of the path
  const int64_t n_tokens = S * L;
  if (n_tokens > 0x7fffffffLL || H > 4096) return 0;
  const int nb = L <= 32 ? 2 : L <= 48 ? 3 : 4;  // 16-row blocks of an item
  const int bwaves = waves_for(nb);
  BwdGeo g;
  g.qkv = qkv; g.dctx = dctx; g.kpm = kpm; g.dqkv = dqkv;
  g.n_tokens = (int)n_tokens; g.L = (int)L; g.H = (int)H;
  g.scale = 0.125f;  // 1 / sqrt(64)
  g.dr = dr; g.site = site;
  g.grp_ptr = nullptr; g.seg_start = nullptr;
  if (L <= 16) { g.P = (int)(16 / L); g.rows_per_item = 2 * g.P * (int)L; }
  else { g.P = 1; g.rows_per_item = (int)L; }
  const int64_t items = (n_tokens + g.rows_per_item - 1) / g.rows_per_item;
  if (items > 0x7fffffffLL) return 0;
  g.n_items = (int)items;
  // one workgroup (4 / 3 / 2 waves, 128-148 KB of LDS) per CU; waves = chunks x heads, at most 256 chunks (the slab scratch)
  int64_t chunks = ((int64_t)stlt_device_cus() * bwaves) / H;
  if (chunks < 1) chunks = 1;
  if (chunks > items) chunks = items;
  if (chunks > 256) chunks = 256;
  g.chunks = (int)chunks;
  g.cs = want_colsum ? scratch : nullptr;
  if (chunks_out) *chunks_out = (int)chunks;
  const int n_wg = (int)((chunks * H + bwaves - 1) / bwaves);
  int rc;
  if (L <= 16) rc = launch_bwd16<2, false, false>(g, n_wg, s);
  else if (nb == 2) rc = causal ? launch_bwd16<2, true, true>(g, n_wg, s) : launch_bwd16<2, true, false>(g, n_wg, s);
  else if (nb == 3) rc = causal ? launch_bwd16<3, true, true>(g, n_wg, s) : launch_bwd16<3, true, false>(g, n_wg, s);
  else rc = causal ? launch_bwd16<4, true, true>(g, n_wg, s) : launch_bwd16<4, true, false>(g, n_wg, s);
  *taken = true;
  return rc;
}
