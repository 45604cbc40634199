// K3 — attention core: ctx = softmax(Q·Kᵀ/sqrt(dh) + M)·V per (sequence, head), masks generated in-kernel
// from the key-padding bytes and (temporal) j>i; the (L,L) mask of utils/model_utils.py:4-7 is never built.
//
// Token space: the packed QKV buffer is (S*L, 3*H*dh); sequences are contiguous runs of L tokens.  A
// wavefront owns 32 consecutive query tokens of one head.  Short sequences (L <= 16, the per-frame object
// sequences, L = 7 at cfg2) are packed P = floor(32/L) to a tile and separated by a block-diagonal mask;
// longer ones (temporal L = 32/33/64...) are walked in 32-key tiles with an online softmax.
//
// Per tile the wave computes Sᵀ = K·Qᵀ (swapped operands) with v_mfma_f32_32x32x2_f32: lane (i = lane&31,
// h = lane>>5) then holds query i's scores for 16 of the 32 keys in its accumulator registers, so the row
// max / row sum are in-register plus one cross-half shuffle, and the probabilities are already the
// B operand of Oᵀ += Vᵀ·Pᵀ (MFMA sums over the accumulator's row index: no lane movement, no LDS round
// trip).  Q/K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: fully coalesced 256-B head rows, no
// staging VGPRs), XOR-swizzled for conflict-free fragment reads; the output tile leaves through LDS as whole rows.
#include <cstdio>
#include <cstdlib>
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int TILE = 32;
constexpr int TILE_FLOATS = TILE * DH;  // 8 KB, unpadded: LDS-DMA writes 1 KB (4 rows) per wave instruction

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// LDS image of a 32x64 tile: row r, 16-B chunk q lives at chunk slot q ^ (r & 15).  With 256-B rows every row
// starts on bank 0, so an unswizzled ds_read_b128 column read would be a 16-way conflict; the XOR spreads the 16
// lanes of each ds_read_b128 group over all 16 slots.  LDS-DMA writes lane-linearly, so the swizzle is applied
// to the per-lane SOURCE address (same 256-B row -> coalescing unchanged) and again on every read.
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ (row & 15); }

// 8 x global_load_lds_dwordx4: no VGPR staging, no ds_write; rows past the end re-read the last valid row
// (their keys are masked / their query rows never stored).
__device__ __forceinline__ void tile_dma(float* lds_tile, const float* __restrict__ src, int64_t ld, int rows_valid,
                                         int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + (lane >> 4), slot = lane & 15;
    const int rr = row < rows_valid ? row : rows_valid - 1;
    const float* g = src + (int64_t)rr * ld + swz(row, slot) * 4;
    __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(lds_tile + i * 256), 16, 0, 0);
  }
}

// MFMA A/B operand fragments of a 32x64 tile straight from global memory: lane (r, h) holds, for c = 0..7, the
// 4 consecutive channels of 16-B chunk 2c+h of row r (rows past the end re-read the last valid row).
__device__ __forceinline__ void frag_load(f32x4 (&f)[8], const float* __restrict__ src, int64_t ld, int rows_valid,
                                          int li, int lh) {
  const int rr = li < rows_valid ? li : rows_valid - 1;
  const float* row = src + (int64_t)rr * ld + 4 * lh;
#pragma unroll
  for (int c = 0; c < 8; ++c) f[c] = *reinterpret_cast<const f32x4*>(row + 8 * c);
}

// A wave only ever touches its own LDS slice, so synchronisation is wave-local: wait for this wave's LDS-DMA
// (vmcnt) and LDS ops (lgkmcnt); DS operations of one wave execute in order.  No s_barrier: the waves of a
// workgroup walk different items.
__device__ __forceinline__ void wave_mem_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

constexpr int WAVES = 4;                                   // independent waves per workgroup
constexpr int WAVE_LDS_FLOATS = 2 * TILE_FLOATS + TILE;    // two V buffers + key metadata

// Geometry of one step = (item, key tile).  item = (token group, query tile, head), head fastest so that the
// waves of a workgroup read adjacent 256-B head slices of the same rows.
struct StepGeo {
  int head, qb, q_first, q_rows, k_first, k_rows, kt_end;
  int64_t q_tok0, k_tok0;
};

// Problem geometry.  Self-attention on a packed (tokens, 3d) buffer is q = qkv, k = qkv + d, v = qkv + 2d,
// ldq = ldkv = 3d, Lq = Lk.  Cross-attention (CAF, reference models.py:362-382) has its own key/value token space.
struct AttnGeo {
  const float* q; const float* k; const float* v;
  int64_t ldq, ldkv;             // row strides in floats
  int64_t nq_tokens, nk_tokens;  // S*Lq, S*Lk
  int Lq, Lk, GLq, GLk, ntq, ntk, H, causal;
  // Ragged mode (VARLEN kernels): self-attention over nq_tokens compacted rows cut into variable-length segments;
  // seg_start[row] / seg_end[row] = first row / one past the last row of the row's segment.  A query tile is 32
  // consecutive rows whatever the segment boundaries; its keys are the rows from its first query's segment start to
  // its last query's segment end, walked in 32-row tiles; the mask is "same segment" (+ key row <= query row).
  const int* seg_start; const int* seg_end;
  // Walk the items from the last to the first: the producer of the packed QKV buffer (the in-projection GEMM) wrote
  // its rows in ascending order, so the tail of the buffer is what the memory-side cache still holds.
  int64_t n_items; int reverse;
};

// 32-bit index arithmetic throughout (the launchers bound items and tokens by 2^31): the 64-bit divisions this replaced
// were a visible part of a wave's first item, which is all a small launch consists of.
template <bool VARLEN>
__device__ __forceinline__ StepGeo step_geo(int64_t item64, int kt, const AttnGeo& a) {
  StepGeo s;
  unsigned item = (unsigned)item64;
  if (a.reverse) item = (unsigned)a.n_items - 1u - item;
  const unsigned H = (unsigned)a.H;
  const unsigned tile_id = item / H;
  s.head = (int)(item - tile_id * H);
  if (VARLEN) {
    s.qb = 0;
    s.q_tok0 = 0;
    s.k_tok0 = 0;
    s.q_first = (int)tile_id * TILE;
    const int left = (int)a.nq_tokens - s.q_first;
    s.q_rows = left < TILE ? left : TILE;
    const int klo = a.seg_start[s.q_first];
    const int khi = a.causal ? s.q_first + s.q_rows : a.seg_end[s.q_first + s.q_rows - 1];
    s.k_first = klo + kt * TILE;
    s.k_rows = khi - s.k_first < TILE ? khi - s.k_first : TILE;
    s.kt_end = (khi - klo + TILE - 1) / TILE;
    return s;
  }
  const unsigned g = tile_id / (unsigned)a.ntq;
  s.qb = (int)(tile_id - g * (unsigned)a.ntq);
  s.q_tok0 = (int64_t)(g * (unsigned)a.GLq);
  s.k_tok0 = (int64_t)(g * (unsigned)a.GLk);
  const int left_q = (int)a.nq_tokens - (int)s.q_tok0, left_k = (int)a.nk_tokens - (int)s.k_tok0;
  const int gq = left_q < a.GLq ? left_q : a.GLq;
  const int gk = left_k < a.GLk ? left_k : a.GLk;
  s.q_first = s.qb * TILE;
  s.q_rows = gq - s.q_first < TILE ? gq - s.q_first : TILE;
  s.k_first = kt * TILE;
  s.k_rows = gk - s.k_first < TILE ? gk - s.k_first : TILE;
  s.kt_end = (a.causal && a.GLq == a.Lq) ? s.qb + 1 : a.ntk;  // causal: key tiles past the query tile are fully masked
  return s;
}

// Default on (measured at cfg2, 1024 clips: temporal 89.6 -> 79.1 us per launch, 0.56 -> 0.64 of HBM peak; the 302-MB
// QKV buffer is larger than the 256-MB memory-side cache, so walking it oldest-first misses everywhere while
// newest-first finds most of it still cached).  STLT_ATTN_REVERSE=0 restores ascending order for A/B measurements.
inline int attn_reverse_order() {
  static int rev = -1;
  if (rev < 0) { const char* e = getenv("STLT_ATTN_REVERSE"); rev = e ? (atoi(e) != 0) : 1; }
  return rev;
}

// CAUSAL is also the kernel's name tag: the temporal pass (causal + key padding) and the spatial pass (key padding only)
// are distinct symbols, so a kernel trace reports each one's launches and durations on its own.
template <bool STAMP, bool VARLEN, bool CAUSAL>
__global__ __launch_bounds__(64 * WAVES, 2) void attn_core_kernel(const AttnGeo geo,
                                                                  const uint8_t* __restrict__ kpm,
                                                                  int64_t n_items, float scale,
                                                                  float* __restrict__ ctx,
                                                                  unsigned long long* __restrict__ stamps, StltDrop dr,
                                                                  uint32_t site) {
  const int H = geo.H;
  constexpr int causal = CAUSAL ? 1 : 0;
  const float* __restrict__ gq = geo.q;
  const float* __restrict__ gk = geo.k;
  const float* __restrict__ gv = geo.v;
  const int64_t ldq = geo.ldq, ldkv = geo.ldkv;
  __shared__ __attribute__((aligned(16))) float smem_all[WAVES * WAVE_LDS_FLOATS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* smem = smem_all + wave * WAVE_LDS_FLOATS;
  int* kmeta = reinterpret_cast<int*>(smem + 2 * TILE_FLOATS);

  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int d = H * DH;
  const int64_t stride = (int64_t)gridDim.x * WAVES;  // persistent waves: item, item + stride, ...
  int64_t item = (int64_t)blockIdx.x * WAVES + wave;
  if (item >= n_items) return;

  unsigned long long ts[8];
#define STAMP_AT(k) do { if (STAMP) { __builtin_amdgcn_sched_barrier(0); ts[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } } while (0)
  STAMP_AT(0);

  f32x4 qf[8], kf[8];
  int kt = 0, buf = 0;
  int pad;    // fixed-L: the key's padding byte; ragged: the key row's segment start
  int qseg = 0;  // ragged: this lane's query row's segment start
  {  // prologue: first step's loads
    const StepGeo s0 = step_geo<VARLEN>(item, 0, geo);
    frag_load(qf, gq + (s0.q_tok0 + s0.q_first) * ldq + s0.head * DH, ldq, s0.q_rows, li, lh);
    frag_load(kf, gk + (s0.k_tok0 + s0.k_first) * ldkv + s0.head * DH, ldkv, s0.k_rows, li, lh);
    tile_dma(smem, gv + (s0.k_tok0 + s0.k_first) * ldkv + s0.head * DH, ldkv, s0.k_rows, lane);
    if (VARLEN) {
      pad = geo.seg_start[s0.k_first + (li < s0.k_rows ? li : 0)];
      qseg = geo.seg_start[s0.q_first + (li < s0.q_rows ? li : 0)];
    } else {
      pad = kpm[s0.k_tok0 + s0.k_first + (li < s0.k_rows ? li : 0)];
    }
  }

  f32x16 o0, o1;
  float m_run = -1e30f, l_run = 0.f;

  for (;;) {
    const StepGeo s = step_geo<VARLEN>(item, kt, geo);
    float* Vs = smem + buf * TILE_FLOATS;
    STAMP_AT(1);
    wave_mem_sync();  // this step's fragments + V tile have landed (and the previous item's stores retired)
    STAMP_AT(2);
    {
      // key metadata for the mask: -1 = masked/absent, else (sequence id << 16) | position in sequence
      if (VARLEN) {
        if (lane < TILE) kmeta[lane] = li < s.k_rows ? pad : -1;  // segment start of the key row
      } else {
        const int kj = s.k_first + li;
        const int ks = kj / geo.Lk;
        if (lane < TILE) kmeta[lane] = (li < s.k_rows && pad == 0) ? ((ks << 16) | (kj - ks * geo.Lk)) : -1;
      }
    }
    if (kt == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
      m_run = -1e30f;
      l_run = 0.f;
    }

    // Sᵀ[j][i] = sum_k K[j][k] Q[i][k]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c][e], qf[c][e], st, 0, 0, 0);
    }

    // The K (and, at the end of an item, Q) registers are dead: put the next step's loads in flight now so
    // that mask/softmax, P·V, the epilogue and the stores run under their latency.
    int64_t n_item = item;
    int n_kt = kt + 1;
    if (n_kt >= s.kt_end) { n_item = item + stride; n_kt = 0; }
    const bool have_next = n_item < n_items;
    int n_pad = 0, n_qseg = qseg;
    if (have_next) {
      const StepGeo sn = step_geo<VARLEN>(n_item, n_kt, geo);
      if (n_kt == 0) frag_load(qf, gq + (sn.q_tok0 + sn.q_first) * ldq + sn.head * DH, ldq, sn.q_rows, li, lh);
      frag_load(kf, gk + (sn.k_tok0 + sn.k_first) * ldkv + sn.head * DH, ldkv, sn.k_rows, li, lh);
      tile_dma(smem + (buf ^ 1) * TILE_FLOATS, gv + (sn.k_tok0 + sn.k_first) * ldkv + sn.head * DH, ldkv, sn.k_rows, lane);
      if (VARLEN) {
        n_pad = geo.seg_start[sn.k_first + (li < sn.k_rows ? li : 0)];
        if (n_kt == 0) n_qseg = geo.seg_start[sn.q_first + (li < sn.q_rows ? li : 0)];
      } else {
        n_pad = kpm[sn.k_tok0 + sn.k_first + (li < sn.k_rows ? li : 0)];
      }
    }
    if (STAMP) asm volatile("" :: "v"(st[0]), "v"(st[15]));
    STAMP_AT(3);

    // mask + online softmax; register r <-> key j = (r&3) + 8*(r>>2) + 4*lh.  Branch-free: all 16 metadata
    // words are fetched first (two addresses per instruction -> LDS broadcast).
    const int qi = s.q_first + li;
    const int q_seq = qi / geo.Lq, q_pos = qi - q_seq * geo.Lq;
    const int q_hi = q_seq << 16;
    const int pos_lim = causal ? q_pos : 0xffff;
    int meta[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) meta[r] = kmeta[(r & 3) + 8 * (r >> 2) + 4 * lh];
    float p[16];
    float m_tile = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // valid key: meta >= 0, same sequence, position <= limit
      const bool ok = VARLEN ? ((meta[r] == qseg) & (s.k_first + (r & 3) + 8 * (r >> 2) + 4 * lh <= (causal ? qi : 0x7fffffff)))
                             : ((meta[r] >= 0) & ((meta[r] & ~0xffff) == q_hi) & ((meta[r] & 0xffff) <= pos_lim));
      p[r] = ok ? st[r] * scale : -1e30f;
      m_tile = fmaxf(m_tile, p[r]);
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_new = fmaxf(m_run, m_tile);
    const float alpha = __expf(m_run - m_new);  // 1 when nothing changed, 0 on the first unmasked tile
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = p[r] > -1e29f ? __expf(p[r] - m_new) : 0.f;
      psum += p[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (kt > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    if (STAMP) asm volatile("" :: "v"(p[0]), "v"(p[15]), "v"(l_run));
    STAMP_AT(4);

    // Oᵀ[c][i] += sum_j V[j][c] P[j][i] ; MFMA step r sums keys j(r,0) and j(r,1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float* vrow = Vs + j * DH + (li & 3);
      const float v0 = vrow[swz(j, li >> 2) * 4];
      const float v1 = vrow[swz(j, 8 + (li >> 2)) * 4];
      float pr = p[r];
      if (dr.thr) {  // train-mode dropout of the attention probabilities: the denominator keeps the undropped sum
        // key position inside its sequence: fixed-L metadata carries it; ragged metadata is the segment's first row
        const uint64_t kpos = VARLEN ? (uint64_t)((s.k_first + j - meta[r]) & 0xff) : (uint64_t)(meta[r] & 0xff);
        const uint64_t idx = ((((uint64_t)(s.q_tok0 + qi)) * H + s.head) << 8) | kpos;
        pr = stlt_keep(dr, site, idx) ? pr * dr.scale : 0.f;
      }
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, pr, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, pr, o1, 0, 0, 0);
    }
    if (STAMP) asm volatile("" :: "v"(o0[0]), "v"(o1[15]));
    STAMP_AT(5);

    if (kt + 1 >= s.kt_end) {
      const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
      const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;  // fully masked row -> zeros
      // Epilogue through LDS (this step's V buffer is dead after the last P·V) so that every global store
      // instruction writes whole 256-B head rows.  lane (i,h), registers 4q..4q+3 <-> channels 8q+4h+(0..3)
      // (+32 for o1).  DS ops of a wave execute in order: no wait needed between the P·V reads and these writes.
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 a = {o0[4 * q] * inv, o0[4 * q + 1] * inv, o0[4 * q + 2] * inv, o0[4 * q + 3] * inv};
        f32x4 b = {o1[4 * q] * inv, o1[4 * q + 1] * inv, o1[4 * q + 2] * inv, o1[4 * q + 3] * inv};
        *reinterpret_cast<f32x4*>(Vs + li * DH + swz(li, 2 * q + lh) * 4) = a;
        *reinterpret_cast<f32x4*>(Vs + li * DH + swz(li, 8 + 2 * q + lh) * 4) = b;
      }
      STAMP_AT(6);
      float* obase = ctx + (s.q_tok0 + s.q_first) * (int64_t)d + s.head * DH;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lane >> 4), chunk = lane & 15;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Vs + row * DH + swz(row, chunk) * 4);
        if (row < s.q_rows) *reinterpret_cast<f32x4*>(obase + (int64_t)row * d + chunk * 4) = v;
      }
      if (STAMP) {
        STAMP_AT(7);
        if (lane == 0 && stamps) {
          for (int k = 0; k < 8; ++k) stamps[item * 8 + k] = ts[k];
        }
        STAMP_AT(0);
      }
    }
    if (!have_next) break;
    item = n_item;
    kt = n_kt;
    pad = n_pad;
    qseg = n_qseg;
    buf ^= 1;
  }
#undef STAMP_AT
}

}  // namespace


// General entry: queries (S*Lq rows of q, stride ldq) attend to keys/values (S*Lk rows of k / v, stride ldkv).
int launch_attn_general(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kpm,
                        int causal, int64_t S, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float* ctx, int kid,
                        hipStream_t s, StltDrop dr, uint32_t site) {
  if (!q || !k || !v || !kpm || !ctx) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: null pointer");
  if (dh != DH) {  // any other head dim: the vector-ALU kernels of attn_any.hip
    if (causal && Lq != Lk) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: causal masking needs Lq == Lk");
    StltProfScope ps(kid, s);
    return launch_attn_any_fwd(q, ldq, k, v, ldkv, kpm, nullptr, nullptr, causal, S * Lq, Lq, Lk, H, dh, ctx, s, dr, site);
  }
  if (dr.thr && Lk > 256) return stlt_set_error(STLT_EINVAL, "attention dropout supports sequences of at most 256 tokens");
  if (Lq <= 0 || Lk <= 0 || Lq > 32768 || Lk > 32768 || H <= 0 || H > 65535) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: bad L/H");
  if (causal && Lq != Lk) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: causal masking needs Lq == Lk");
  if (ldq % 4 || ldkv % 4) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: strides must be multiples of 4 floats");
  if (S == 0) return 0;
  // sequences packed per 32-token tile (short self-attention sequences only)
  const int P = (Lq == Lk && Lq <= 16) ? (int)(TILE / Lq) : 1;
  AttnGeo g;
  g.q = q; g.k = k; g.v = v; g.ldq = ldq; g.ldkv = ldkv;
  g.nq_tokens = S * Lq; g.nk_tokens = S * Lk;
  g.Lq = (int)Lq; g.Lk = (int)Lk; g.GLq = P * (int)Lq; g.GLk = P * (int)Lk;
  g.ntq = (g.GLq + TILE - 1) / TILE; g.ntk = (g.GLk + TILE - 1) / TILE;
  g.H = (int)H; g.causal = causal;
  g.seg_start = nullptr; g.seg_end = nullptr;
  const int64_t groups = (S + P - 1) / P;
  if (groups * g.ntq * H > 0x7fffffffLL || g.nq_tokens > 0x7fffffffLL || g.nk_tokens > 0x7fffffffLL)
    return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: too many tiles / tokens (32-bit index arithmetic)");
  StltProfScope ps(kid, s);
  stlt_prof_note("attn S=%lld Lq=%lld Lk=%lld H=%lld causal=%d", (long long)S, (long long)Lq, (long long)Lk, (long long)H, causal);
  stlt_prof_add_bytes(4.0 * (double)H * DH * (2.0 * (double)(S * Lq) + 2.0 * (double)(S * Lk)) + (double)(S * Lk));  // read q, k, v; write ctx; one mask byte per key
  stlt_prof_note_flops(4.0 * (double)S * (double)H * (double)Lq * (double)Lk * DH);
  // short self-attention on a packed buffer (with or without the dropout of the probabilities): the 16-row-tile kernel (attn16.hip)
  if (Lq == Lk && Lq <= 64 && k == q + H * dh && v == q + 2 * H * dh && ldq == 3 * H * dh && ldkv == ldq && !g_stlt_debug_buf) {
    bool taken = false;
    const int rc = launch_attn16(q, kpm, causal, S, Lq, H, ctx, attn_reverse_order(), s, &taken, dr, site);
    if (taken || rc != 0) return rc;
  }
  const int64_t n_items = groups * g.ntq * H;
  g.n_items = n_items;
  g.reverse = attn_reverse_order();
  // persistent waves: as many workgroups as are resident at once (occupancy API x CU count), each wave strides
  // over the items; there is no inter-workgroup dependency, so a wrong residency guess only costs speed
  const int n_cu = stlt_device_cus();
  static StltPerDeviceInt occ;  // resident workgroups per CU, per device (the spatial and temporal instantiations use the same resources)
  int& wg_per_cu = occ.ref();
  if (wg_per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, attn_core_kernel<false, false, false>, 64 * WAVES, 0) != hipSuccess || wg_per_cu <= 0)
      wg_per_cu = 1;
    if (getenv("STLT_DEBUG")) fprintf(stderr, "[stlt] attn: device %d: %d CUs, %d workgroups/CU of %d waves\n", stlt_current_device(), n_cu, wg_per_cu, WAVES);
  }
  int64_t n_wg = (n_items + WAVES - 1) / WAVES;
  if (n_wg > (int64_t)wg_per_cu * n_cu) n_wg = (int64_t)wg_per_cu * n_cu;
  dim3 grid((unsigned)n_wg);
  const float scale = 1.0f / sqrtf((float)dh);
  if (g_stlt_debug_buf && !getenv("STLT_GEMM_STAMP")) {  // diagnostic build path only (tools/attn_stamps.py); never set by the product
    if (causal) hipLaunchKernelGGL((attn_core_kernel<true, false, true>), grid, dim3(64 * WAVES), 0, s, g, kpm, n_items, scale, ctx, g_stlt_debug_buf, dr, site);
    else hipLaunchKernelGGL((attn_core_kernel<true, false, false>), grid, dim3(64 * WAVES), 0, s, g, kpm, n_items, scale, ctx, g_stlt_debug_buf, dr, site);
  } else if (causal) {
    hipLaunchKernelGGL((attn_core_kernel<false, false, true>), grid, dim3(64 * WAVES), 0, s, g, kpm, n_items, scale, ctx, (unsigned long long*)nullptr, dr, site);
  } else {
    hipLaunchKernelGGL((attn_core_kernel<false, false, false>), grid, dim3(64 * WAVES), 0, s, g, kpm, n_items, scale, ctx, (unsigned long long*)nullptr, dr, site);
  }
  return stlt_check_launch("attn_core_kernel");
}

int launch_attn(const float* qkv, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, int64_t dh,
                float* ctx, int kid, hipStream_t s, StltDrop dr, uint32_t site) {
  if (!qkv) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: null pointer");
  const int64_t d = H * dh;
  return launch_attn_general(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, kpm, causal, S, L, L, H, dh, ctx, kid, s, dr, site);
}

// Ragged self-attention over M compacted rows of a packed (M, 3*H*dh) buffer (see AttnGeo): every row is a real
// token, so there is no key-padding mask.
int launch_attn_ragged(const float* qkv, const int* seg_start, const int* seg_end, int causal, int64_t M, int64_t H, int64_t dh,
                       float* ctx, int kid, hipStream_t s, StltDrop dr, uint32_t site) {
  if (!qkv || !seg_start || !seg_end || !ctx) return stlt_set_error(STLT_EINVAL, "attn_ragged: null pointer");
  if (dh != DH) {
    StltProfScope ps(kid, s);
    const int64_t d = H * dh;
    return launch_attn_any_fwd(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, nullptr, seg_start, seg_end, causal, M, 256, 256, H, dh, ctx, s, dr, site);  // segments: frames / clips; the whole-path callers refuse layouts whose segments could exceed the kernel's 1024 keys (api.hip: forward_ragged)
  }
  if (H <= 0 || H > 65535 || M < 0 || M > 0x7fffff00LL) return stlt_set_error(STLT_EINVAL, "attn_ragged: bad M/H");
  if (M == 0) return 0;
  const int64_t d = H * dh;
  AttnGeo g;
  g.q = qkv; g.k = qkv + d; g.v = qkv + 2 * d; g.ldq = 3 * d; g.ldkv = 3 * d;
  g.nq_tokens = M; g.nk_tokens = M;
  g.Lq = g.Lk = g.GLq = g.GLk = TILE; g.ntq = g.ntk = 1;
  g.H = (int)H; g.causal = causal;
  g.seg_start = seg_start; g.seg_end = seg_end;
  const int64_t n_items = ((M + TILE - 1) / TILE) * H;
  if (n_items > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "attn_ragged: too many tiles");
  g.n_items = n_items;
  g.reverse = attn_reverse_order();
  StltProfScope ps(kid, s);
  const int n_cu = stlt_device_cus();
  static StltPerDeviceInt occ;
  int& wg_per_cu = occ.ref();
  if (wg_per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, attn_core_kernel<false, true, false>, 64 * WAVES, 0) != hipSuccess || wg_per_cu <= 0)
      wg_per_cu = 1;
  }
  int64_t n_wg = (n_items + WAVES - 1) / WAVES;
  if (n_wg > (int64_t)wg_per_cu * n_cu) n_wg = (int64_t)wg_per_cu * n_cu;
  if (causal)
    hipLaunchKernelGGL((attn_core_kernel<false, true, true>), dim3((unsigned)n_wg), dim3(64 * WAVES), 0, s, g, (const uint8_t*)nullptr, n_items,
                       1.0f / sqrtf((float)dh), ctx, (unsigned long long*)nullptr, dr, site);
  else
    hipLaunchKernelGGL((attn_core_kernel<false, true, false>), dim3((unsigned)n_wg), dim3(64 * WAVES), 0, s, g, (const uint8_t*)nullptr, n_items,
                       1.0f / sqrtf((float)dh), ctx, (unsigned long long*)nullptr, dr, site);
  return stlt_check_launch("attn_core_kernel(ragged)");
}
