// Attention backward on v_mfma_f32_16x16x4_f32 for queries and keys from DIFFERENT buffers (the fusion models' cross-attention, reference
// models.py:362-382 and 411-419: 32 layout frames against 33 appearance tokens and the other way round): attn_bwd16.hip's dataflow with a
// query side of NBQ and a key side of NBK sixteen-row blocks, no causal mask, the key-padding mask over the keys, the dropout mask of the
// forward (stlt_attn_fwd_dropout: element index ((query token * H + head) << 8) | key position).
//   P = softmax(scale·Q·Kᵀ + mask), Pd = P∘D, O = Pd·V
//   dV = Pdᵀ·dO ; dPd = dO·Vᵀ ; dP = dPd∘D ; dS = P∘(dP − rowsum(P∘dP)) ; dQ = scale·dS·K ; dK = scale·dSᵀ·Q
// One wave per (sequence, head); Q, dO (16 NBQ x 64) and K, V (16 NBK x 64) tiles by LDS-DMA in attn16.hip's swizzled layout; scores
// and dPd transposed (Sᵀ = K·Qᵀ: a query's row in 4 lanes x 4 registers per key block), dS / Pd through an LDS transpose for dK / dV (in
// the V / K tiles' space once those are dead).  No column sums: the callers' in-projection products take the bias gradient from dq / dk /
// dv.  Replaces bwd_api.hip's attn_bwd_general_kernel (vector ALU, one workgroup per (sequence, head): 72 us for 64 x 12 items of 32 x 33)
// for sequences of at most 48 tokens on either side.
#include <cstdlib>
#include "common.h"
#include "wave_dpp.h"

namespace {

constexpr int BD = 64;
// waves per workgroup: (NBQ + NBK) x 8 KB of LDS per wave
constexpr int xwaves(int nbq, int nbk) { return nbq + nbk <= 4 ? 4 : nbq + nbk <= 6 ? 3 : 2; }

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

struct XGeo {
  const float* q; const float* k; const float* v; const float* dctx; const uint8_t* kpm;
  float* dq; float* dk; float* dv;
  int64_t ldq, ldkv, lddq, lddkv;
  int S, Lq, Lk, H, chunks;
  float scale;
  StltDrop dr;
  uint32_t site;
};

template <int NBQ, int NBK, bool DROP>
__global__ __launch_bounds__(64 * xwaves(NBQ, NBK)) void attn_bwdx16_kernel(const XGeo geo) {
  constexpr int RQ = 16 * NBQ, RK = 16 * NBK, TQ = RQ * BD, TK = RK * BD, W = xwaves(NBQ, NBK);
  static_assert(RQ <= 64 && RK <= 64, "the transposed dS / Pd live in a K / V tile: at most 64 queries");
  __shared__ __attribute__((aligned(16))) float smem_all[W * (2 * TQ + 2 * TK + RK)];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* Qs = smem_all + wave * (2 * TQ + 2 * TK + RK);
  float* Gs = Qs + TQ;
  float* Ks = Gs + TQ;
  float* Vs = Ks + TK;
  int* kmeta = reinterpret_cast<int*>(Vs + TK);
  float* Tds = Vs;  // dS[key][query], RK x RQ, once the V tile is dead
  float* Tpd = Ks;  // Pd[key][query], once the K tile is dead
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4;
  const int H = geo.H, Lq = geo.Lq, Lk = geo.Lk, d = H * BD;
  const int w = blockIdx.x * W + wave;
  const int head = w % H, chunk = w / H;
  if (chunk >= geo.chunks) return;  // the grid is rounded up to whole workgroups
  auto swz = [&](int row, int c) __attribute__((always_inline)) { return row * BD + ((c ^ (row & 15)) << 2); };

  for (int item = chunk; item < geo.S; item += geo.chunks) {
    const int64_t tq0 = (int64_t)item * Lq, tk0 = (int64_t)item * Lk;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous item's LDS reads have returned
#pragma unroll
    for (int i = 0; i < RQ / 4; ++i) {
      const int row = 4 * i + (lane >> 4), slot = lane & 15;
      const int64_t tok = tq0 + (row < Lq ? row : 0);  // absent rows re-read a valid row; they are masked / never stored
      const int ch = head * BD + (slot ^ (row & 15)) * 4;
      __builtin_amdgcn_global_load_lds((glb_void_ptr)(geo.q + tok * geo.ldq + ch), (lds_void_ptr)(Qs + i * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_ptr)(geo.dctx + tok * d + ch), (lds_void_ptr)(Gs + i * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RK / 4; ++i) {
      const int row = 4 * i + (lane >> 4), slot = lane & 15;
      const int64_t tok = tk0 + (row < Lk ? row : 0);
      const int ch = head * BD + (slot ^ (row & 15)) * 4;
      __builtin_amdgcn_global_load_lds((glb_void_ptr)(geo.k + tok * geo.ldkv + ch), (lds_void_ptr)(Ks + i * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_ptr)(geo.v + tok * geo.ldkv + ch), (lds_void_ptr)(Vs + i * 256), 16, 0, 0);
    }
    if (lane < RK) kmeta[lane] = (lane < Lk && (!geo.kpm || geo.kpm[tk0 + lane] == 0)) ? lane : -1;  // key position, -1 = absent / padded
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // wave-local: tiles and key metadata are in LDS

    f32x4 ds[NBQ][NBK], pd[NBQ][NBK];  // dSᵀ and Pdᵀ of (query block, key block): lane = query li, register r = key 4*lg + r
#pragma unroll
    for (int qb = 0; qb < NBQ; ++qb) {
      // ---- Sᵀ = K·Qᵀ and dPdᵀ = V·dOᵀ for this query block
      const int qrow = qb * 16 + li;
      f32x4 qf[4], gf[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qf[c] = *reinterpret_cast<const f32x4*>(Qs + swz(qrow, 4 * c + lg));
        gf[c] = *reinterpret_cast<const f32x4*>(Gs + swz(qrow, 4 * c + lg));
      }
      f32x4 st[NBK], dp[NBK];
#pragma unroll
      for (int kb = 0; kb < NBK; ++kb) {
        st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
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
      const bool q_ok = qrow < Lq;
      const int64_t qtok = tq0 + (q_ok ? qrow : 0);
      int kpos[NBK][4];
      float m = -1e30f;
#pragma unroll
      for (int kb = 0; kb < NBK; ++kb) {
        const int4 km = *reinterpret_cast<const int4*>(kmeta + kb * 16 + 4 * lg);
        const int kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = q_ok & (kmv[r] >= 0);
          kpos[kb][r] = kmv[r] & 0xff;
          st[kb][r] = ok ? st[kb][r] * geo.scale : -1e30f;
          m = fmaxf(m, st[kb][r]);
        }
      }
      m = groups_max(m);
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NBK; ++kb) {
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
      for (int kb = 0; kb < NBK; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[kb][r] * inv;
          float g = dp[kb][r];  // dPd
          float pdv = p;
          if (DROP) {  // the forward multiplied P by the mask before P·V: dP = dPd∘D, and dV sees Pd = P∘D
            const uint64_t idx = ((((uint64_t)qtok) * H + head) << 8) | (uint64_t)kpos[kb][r];
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
      for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[qb][kb][r] = st[kb][r] * (dp[kb][r] - dsum);
      // ---- dQᵀ[channel][query] = Kᵀ·dSᵀ: MFMA step (kb, r) sums keys kb*16 + 4g + r over g
      f32x4 o[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) o[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NBK; ++kb) {
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
      if (q_ok) {
        float* dst = geo.dq + qtok * geo.lddq + head * BD + 4 * lg;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) *reinterpret_cast<f32x4*>(dst + 16 * cb) = o[cb] * geo.scale;
      }
    }

    // ---- transposes: dS and Pd with the key in the row, 4 consecutive queries per 16-byte read (K and V tiles are dead)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int qb = 0; qb < NBQ; ++qb)
#pragma unroll
      for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb * 16 + 4 * lg + r;
          Tds[key * RQ + qb * 16 + li] = ds[qb][kb][r];
          Tpd[key * RQ + qb * 16 + li] = pd[qb][kb][r];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- dKᵀ[channel][key] = scale·Qᵀ·dS and dVᵀ[channel][key] = dOᵀ·Pd: MFMA step (qb, r) sums queries qb*16 + 4g + r
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) {
      const int krow = kb * 16 + li;
      f32x4 ok_[4], ov[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) { ok_[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int qb = 0; qb < NBQ; ++qb) {
        const f32x4 bs = *reinterpret_cast<const f32x4*>(Tds + krow * RQ + qb * 16 + 4 * lg);
        const f32x4 bp = *reinterpret_cast<const f32x4*>(Tpd + krow * RQ + qb * 16 + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = qb * 16 + 4 * lg + r;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            const int off = qi * BD + (((cb * 4 + (li >> 2)) ^ (qi & 15)) << 2) + (li & 3);
            ok_[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[off], bs[r], ok_[cb], 0, 0, 0);
            ov[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[off], bp[r], ov[cb], 0, 0, 0);
          }
        }
      }
      if (krow < Lk) {
        float* dk = geo.dk + (tk0 + krow) * geo.lddkv + head * BD + 4 * lg;
        float* dv = geo.dv + (tk0 + krow) * geo.lddkv + head * BD + 4 * lg;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          *reinterpret_cast<f32x4*>(dk + 16 * cb) = ok_[cb] * geo.scale;
          *reinterpret_cast<f32x4*>(dv + 16 * cb) = ov[cb];
        }
      }
    }
  }
}

template <int NBQ, int NBK>
int launch_x(const XGeo& g, hipStream_t s) {
  constexpr int W = xwaves(NBQ, NBK);
  const int n_wg = (int)(((int64_t)g.chunks * g.H + W - 1) / W);
  if (g.dr.thr) hipLaunchKernelGGL((attn_bwdx16_kernel<NBQ, NBK, true>), dim3((unsigned)n_wg), dim3(64 * W), 0, s, g);
  else hipLaunchKernelGGL((attn_bwdx16_kernel<NBQ, NBK, false>), dim3((unsigned)n_wg), dim3(64 * W), 0, s, g);
  return stlt_check_launch("attn_bwdx16_kernel");
}

template <int NBQ>
int launch_xq(int nbk, const XGeo& g, hipStream_t s) {
  switch (nbk) {
    case 1: return launch_x<NBQ, 1>(g, s);
    case 2: return launch_x<NBQ, 2>(g, s);
    default: return launch_x<NBQ, 3>(g, s);
  }
}

}  // namespace

// *taken = true when the launch was made (return value: 0 or the error), false when the shape is not this kernel's (the caller then
// uses bwd_api.hip's attn_bwd_general_kernel): head dim 64, no causal mask, at most 48 tokens on either side, 16-byte aligned rows.
int launch_attn_bwdx16(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* dctx, const uint8_t* kpm,
                       int64_t S, int64_t Lq, int64_t Lk, int64_t H, float* dq, int64_t lddq, float* dk, float* dv, int64_t lddkv, StltDrop dr,
                       uint32_t site, hipStream_t s, bool* taken) {
  *taken = false;
  static const int enabled = [] { const char* e = getenv("STLT_ATTN_BWDX16"); return e ? atoi(e) : 1; }();
  if (enabled != 1 || Lq < 1 || Lk < 1 || Lq > 48 || Lk > 48 || S < 1 || S > 0x7fffffffLL || H < 1 || H > 4096) return 0;
  if (S * (Lq > Lk ? Lq : Lk) > 0x7fffffffLL) return 0;
  if ((ldq | ldkv | lddq | lddkv) % 4 != 0) return 0;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dctx | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) != 0) return 0;
  const int nbq = (int)((Lq + 15) / 16), nbk = (int)((Lk + 15) / 16);
  XGeo g;
  g.q = q; g.k = k; g.v = v; g.dctx = dctx; g.kpm = kpm; g.dq = dq; g.dk = dk; g.dv = dv;
  g.ldq = ldq; g.ldkv = ldkv; g.lddq = lddq; g.lddkv = lddkv;
  g.S = (int)S; g.Lq = (int)Lq; g.Lk = (int)Lk; g.H = (int)H;
  g.scale = 0.125f;
  g.dr = dr; g.site = site;
  const int w = nbq == 1 ? xwaves(1, nbk) : nbq == 2 ? xwaves(2, nbk) : xwaves(3, nbk);
  int64_t chunks = ((int64_t)stlt_device_cus() * w) / H;  // one workgroup per CU, a wave keeps one head
  if (chunks < 1) chunks = 1;
  if (chunks > S) chunks = S;
  g.chunks = (int)chunks;
  *taken = true;
  switch (nbq) {
    case 1: return launch_xq<1>(nbk, g, s);
    case 2: return launch_xq<2>(nbk, g, s);
    default: return launch_xq<3>(nbk, g, s);
  }
}
