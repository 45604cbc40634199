// Attention core, forward and backward, for head dims other than 64.
//
// The reference builds nn.MultiheadAttention(hidden_size, num_attention_heads) for any hidden_size % num_attention_heads == 0
// (src/modelling/configs.py:92-111, models.py:46-52,118-124); every released checkpoint is 768 / 12, and the MFMA kernels of attn.hip /
// attn16.hip / attn_bwd16.hip / mhsa.hip are laid out for that head dim.  This file is the cold path that keeps the boundary whole:
// the same arithmetic (softmax(q kᵀ / sqrt(dh) + mask)·v with masks generated in-kernel, counter-based probability dropout at the
// same element index ((query token·H + head) << 8 | key position), fully masked rows -> zeros) for 1 <= dh <= 256 on the vector ALU.
// It takes every geometry the dh = 64 kernels take: packed self-attention, cross-attention with separate q / k / v strides, causal
// masks, key-padding masks, ragged segments of a compacted buffer (skip-padding), forward and backward.
//
//   forward : one wave per (query row, head).  Scores: lane = key (keys lane, lane + 64, ...), the query row broadcast from LDS;
//             softmax by wave reductions; output: lane = channel, probabilities broadcast from LDS.  Keys <= 1024.
//   backward: one workgroup of 4 waves per (sequence or ragged group, head), two phases and no atomics (bitwise reproducible):
//             phase 1, a wave per query row: softmax statistics (max, 1 / sum, delta = sum_j P dP) kept in LDS, dS row, dq;
//             phase 2, a wave per key row: P and dS recomputed from the statistics with lane = query, then dk and dv with
//             lane = channel.  Queries and keys <= 256 per sequence (the position table's size, as the dh = 64 backward).
#include <cmath>
#include <cstdint>
#include <initializer_list>
#include "common.h"

namespace {

constexpr int ANY_MAX_DH = 256;
constexpr int ANY_MAX_KEYS = 1024;  // forward
constexpr int ANY_MAX_SEQ = 256;    // backward, either side
constexpr int ANY_WAVES = 4;

struct AnyFwd {
  const float* q; const float* k; const float* v;
  int64_t ldq, ldkv;
  const uint8_t* kpm;                        // padded: one byte per key row (S * Lk), 1 = masked
  const int* seg_start; const int* seg_end;  // ragged: per row of the compacted buffer
  float* ctx;                                // (n_q, H * dh)
  int64_t n_q;
  int Lq, Lk, H, dh, causal;
  float scale;
};

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// LDS traffic of one wave is in order; this keeps the compiler from moving accesses across and drains the counters
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// dot product of a row in LDS with a row in global memory (VEC: both 16-byte aligned, n % 4 == 0)
template <bool VEC>
__device__ __forceinline__ float dot_row(const float* __restrict__ s, const float* __restrict__ g, int n) {
  float acc = 0.f;
  if (VEC) {
    for (int c = 0; c < n; c += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(s + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(g + c);
      acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
  } else {
    for (int c = 0; c < n; ++c) acc += s[c] * g[c];
  }
  return acc;
}

template <bool VEC>
__global__ __launch_bounds__(64 * ANY_WAVES) void attn_any_fwd_kernel(const AnyFwd a, int64_t n_items, StltDrop dr, uint32_t site) {
  __shared__ __attribute__((aligned(16))) float q_lds[ANY_WAVES][ANY_MAX_DH];
  __shared__ float p_lds[ANY_WAVES][ANY_MAX_KEYS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* qs = q_lds[wave];
  float* ps = p_lds[wave];
  const int dh = a.dh, H = a.H;
  const uint64_t drop_key = stlt_drop_key(dr, site);
  for (int64_t item = (int64_t)blockIdx.x * ANY_WAVES + wave; item < n_items; item += (int64_t)gridDim.x * ANY_WAVES) {
    const int64_t m = item / H;  // query row; heads fastest: the waves of a workgroup read adjacent head slices of the same rows
    const int head = (int)(item - m * H);
    int64_t k_base;
    int n_k;
    if (a.seg_start) {
      k_base = a.seg_start[m];
      n_k = (int)((a.causal ? m + 1 : (int64_t)a.seg_end[m]) - k_base);
      n_k = n_k < ANY_MAX_KEYS ? n_k : ANY_MAX_KEYS;  // the host cannot see segment lengths; the model's are frames / clips of <= 256 rows
    } else {
      const int64_t sq = m / a.Lq;
      const int i = (int)(m - sq * a.Lq);
      k_base = sq * a.Lk;
      n_k = a.causal ? i + 1 : a.Lk;
    }
    const float* qrow = a.q + m * a.ldq + (int64_t)head * dh;
    wave_lds_sync();  // the previous item's reads of qs / ps are done
    for (int c = lane; c < dh; c += 64) qs[c] = qrow[c];
    wave_lds_sync();
    float mx = -1e30f;
    for (int j = lane; j < n_k; j += 64) {
      const bool ok = a.kpm ? a.kpm[k_base + j] == 0 : true;
      float sc = -1e30f;
      if (ok) sc = dot_row<VEC>(qs, a.k + (k_base + j) * a.ldkv + (int64_t)head * dh, dh) * a.scale;
      ps[j] = sc;
      mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n_k; j += 64) {  // each lane re-reads what it wrote
      const float sc = ps[j];
      float e = sc > -1e29f ? expf(sc - mx) : 0.f;
      sum += e;  // the denominator keeps the undropped sum
      if (dr.thr) e = stlt_keep_k(dr.thr, drop_key, ((((uint64_t)m) * H + head) << 8) | (uint64_t)(j & 0xff)) ? e * dr.scale : 0.f;
      ps[j] = e;
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    wave_lds_sync();
    float* out = a.ctx + m * ((int64_t)H * dh) + (int64_t)head * dh;
    const float* vbase = a.v + k_base * a.ldkv + (int64_t)head * dh;
    for (int c = lane; c < dh; c += 64) {
      float o = 0.f;
      for (int j = 0; j < n_k; ++j) o += ps[j] * vbase[(int64_t)j * a.ldkv + c];
      out[c] = o * inv;
    }
  }
}

struct AnyBwd {
  const float* q; const float* k; const float* v; const float* dctx;  // dctx: (n_q, H * dh)
  int64_t ldq, ldkv;
  float* dq; float* dk; float* dv;
  int64_t lddq, lddkv;
  const uint8_t* kpm;
  const int* grp_ptr; const int* seg_start; const int* seg_end;  // ragged: group g = rows [grp_ptr[g], grp_ptr[g + 1]), q and k in the same buffer
  int Lq, Lk, H, dh, causal;
  float scale;
};

template <bool VEC>
__global__ __launch_bounds__(64 * ANY_WAVES) void attn_any_bwd_kernel(const AnyBwd a, StltDrop dr, uint32_t site) {
  __shared__ float stat_m[ANY_MAX_SEQ], stat_inv[ANY_MAX_SEQ], stat_delta[ANY_MAX_SEQ];
  __shared__ __attribute__((aligned(16))) float row_a[ANY_WAVES][ANY_MAX_DH], row_b[ANY_WAVES][ANY_MAX_DH];
  __shared__ float buf_a[ANY_WAVES][ANY_MAX_SEQ], buf_b[ANY_WAVES][ANY_MAX_SEQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H, dh = a.dh;
  const int64_t g = blockIdx.x / H;
  const int head = (int)(blockIdx.x - g * H);
  int64_t q0, k0;
  int nq, nk;
  if (a.grp_ptr) {
    q0 = k0 = a.grp_ptr[g];
    nq = nk = a.grp_ptr[g + 1] - a.grp_ptr[g];
  } else {
    q0 = g * a.Lq; nq = a.Lq;
    k0 = g * a.Lk; nk = a.Lk;
  }
  const uint64_t drop_key = stlt_drop_key(dr, site);
  const int64_t hoff = (int64_t)head * dh;
  const int64_t ldd = (int64_t)H * dh;
  // key j (local) is visible to query i (local)
  auto visible = [&](int i, int j) -> bool {
    if (a.causal && j > i) return false;
    if (a.seg_start) { const int64_t kr = k0 + j; return kr >= a.seg_start[q0 + i] && kr < a.seg_end[q0 + i]; }
    return a.kpm ? a.kpm[k0 + j] == 0 : true;
  };
  auto key_pos = [&](int i, int j) -> uint64_t {  // position of the key inside the query's sequence (the forward's element index)
    return (uint64_t)((a.seg_start ? (int)(k0 + j - a.seg_start[q0 + i]) : j) & 0xff);
  };
  float* ra = row_a[wave];
  float* rb = row_b[wave];
  float* ba = buf_a[wave];
  float* bb = buf_b[wave];
  // ---- phase 1: statistics, dS rows, dq
  for (int i = wave; i < nq; i += ANY_WAVES) {
    wave_lds_sync();
    for (int c = lane; c < dh; c += 64) {
      ra[c] = a.q[(q0 + i) * a.ldq + hoff + c];
      rb[c] = a.dctx[(q0 + i) * ldd + hoff + c];
    }
    wave_lds_sync();
    float sc[ANY_MAX_SEQ / 64], dp[ANY_MAX_SEQ / 64];
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < ANY_MAX_SEQ / 64; ++t) {
      const int j = lane + 64 * t;
      sc[t] = -1e30f;
      dp[t] = 0.f;
      if (j < nk && visible(i, j)) {
        sc[t] = dot_row<VEC>(ra, a.k + (k0 + j) * a.ldkv + hoff, dh) * a.scale;
        dp[t] = dot_row<VEC>(rb, a.v + (k0 + j) * a.ldkv + hoff, dh);
        if (dr.thr) dp[t] = stlt_keep_k(dr.thr, drop_key, ((((uint64_t)(q0 + i)) * H + head) << 8) | key_pos(i, j)) ? dp[t] * dr.scale : 0.f;
      }
      mx = fmaxf(mx, sc[t]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < ANY_MAX_SEQ / 64; ++t) {
      sc[t] = sc[t] > -1e29f ? expf(sc[t] - mx) : 0.f;
      sum += sc[t];
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < ANY_MAX_SEQ / 64; ++t) {
      sc[t] *= inv;
      delta += sc[t] * dp[t];
    }
    delta = wave_sum(delta);
#pragma unroll
    for (int t = 0; t < ANY_MAX_SEQ / 64; ++t) {
      const int j = lane + 64 * t;
      if (j < nk) ba[j] = sc[t] * (dp[t] - delta);
    }
    if (lane == 0) { stat_m[i] = mx; stat_inv[i] = inv; stat_delta[i] = delta; }
    wave_lds_sync();
    for (int c = lane; c < dh; c += 64) {
      float acc = 0.f;
      for (int j = 0; j < nk; ++j) acc += ba[j] * a.k[(k0 + j) * a.ldkv + hoff + c];
      a.dq[(q0 + i) * a.lddq + hoff + c] = acc * a.scale;
    }
  }
  __syncthreads();
  // ---- phase 2: dk, dv
  for (int j = wave; j < nk; j += ANY_WAVES) {
    wave_lds_sync();
    for (int c = lane; c < dh; c += 64) {
      ra[c] = a.k[(k0 + j) * a.ldkv + hoff + c];
      rb[c] = a.v[(k0 + j) * a.ldkv + hoff + c];
    }
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < ANY_MAX_SEQ / 64; ++t) {
      const int i = lane + 64 * t;
      if (i < nq) {
        float ds = 0.f, pd = 0.f;
        if (visible(i, j)) {
          const float s = dot_row<VEC>(ra, a.q + (q0 + i) * a.ldq + hoff, dh) * a.scale;
          float dpv = dot_row<VEC>(rb, a.dctx + (q0 + i) * ldd + hoff, dh);
          const float p = expf(s - stat_m[i]) * stat_inv[i];
          pd = p;
          if (dr.thr) {
            const bool keep = stlt_keep_k(dr.thr, drop_key, ((((uint64_t)(q0 + i)) * H + head) << 8) | key_pos(i, j));
            pd = keep ? p * dr.scale : 0.f;
            dpv = keep ? dpv * dr.scale : 0.f;
          }
          ds = p * (dpv - stat_delta[i]);
        }
        ba[i] = ds;
        bb[i] = pd;
      }
    }
    wave_lds_sync();
    for (int c = lane; c < dh; c += 64) {
      float ak = 0.f, av = 0.f;
      for (int i = 0; i < nq; ++i) {
        ak += ba[i] * a.q[(q0 + i) * a.ldq + hoff + c];
        av += bb[i] * a.dctx[(q0 + i) * ldd + hoff + c];
      }
      a.dk[(k0 + j) * a.lddkv + hoff + c] = ak * a.scale;
      a.dv[(k0 + j) * a.lddkv + hoff + c] = av;
    }
  }
}

bool rows_vectorise(int64_t dh, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
  if (dh % 4) return false;
  for (int64_t ld : lds) if (ld % 4) return false;
  for (const void* p : ptrs) if ((uintptr_t)p & 15) return false;
  return true;
}

}  // namespace

// Forward.  Padded: n_q = S * Lq query rows against S * Lk key rows, kpm over the key rows.  Ragged (seg_start != null): n_q rows of a
// compacted buffer that holds queries and keys alike; Lq / Lk = the longest possible segment.
int launch_attn_any_fwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kpm, const int* seg_start,
                        const int* seg_end, int causal, int64_t n_q, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float* ctx, hipStream_t s,
                        StltDrop dr, uint32_t site) {
  if (!q || !k || !v || !ctx || (!kpm && !seg_start) || (seg_start && !seg_end)) return stlt_set_error(STLT_EINVAL, "attention (head dim %lld): null pointer", (long long)dh);
  if (dh < 1 || dh > ANY_MAX_DH) return stlt_set_error(STLT_EINVAL, "attention: head dim %lld unsupported (1 ... %d)", (long long)dh, ANY_MAX_DH);
  if (Lq <= 0 || Lk <= 0 || Lk > ANY_MAX_KEYS)
    return stlt_set_error(STLT_EINVAL, "attention with head dim %lld (not 64) takes at most %d keys per sequence (got %lld)", (long long)dh, ANY_MAX_KEYS, (long long)Lk);
  if (dr.thr && Lk > 256) return stlt_set_error(STLT_EINVAL, "attention dropout supports sequences of at most 256 tokens");
  if (H <= 0 || H > 65535 || n_q < 0 || n_q > 0x7fffff00LL) return stlt_set_error(STLT_EINVAL, "attention: bad row / head count");
  if (n_q == 0) return 0;
  AnyFwd a{q, k, v, ldq, ldkv, seg_start ? nullptr : kpm, seg_start, seg_end, ctx, n_q, (int)Lq, (int)Lk, (int)H, (int)dh, causal, 1.0f / sqrtf((float)dh)};
  const int64_t n_items = n_q * H;
  int64_t n_wg = (n_items + ANY_WAVES - 1) / ANY_WAVES;
  const int64_t cap = (int64_t)stlt_device_cus() * 8;
  if (n_wg > cap) n_wg = cap;
  if (rows_vectorise(dh, {ldq, ldkv}, {q, k, v}))
    hipLaunchKernelGGL((attn_any_fwd_kernel<true>), dim3((unsigned)n_wg), dim3(64 * ANY_WAVES), 0, s, a, n_items, dr, site);
  else
    hipLaunchKernelGGL((attn_any_fwd_kernel<false>), dim3((unsigned)n_wg), dim3(64 * ANY_WAVES), 0, s, a, n_items, dr, site);
  return stlt_check_launch("attn_any_fwd_kernel");
}

// Backward.  Padded: n_groups sequences of Lq queries / Lk keys.  Ragged (grp_ptr != null): n_groups row ranges of one compacted
// buffer, each a whole number of segments (seg_start / seg_end per row; null = a group is one segment), at most max_rows rows.
int launch_attn_any_bwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* dctx, const uint8_t* kpm,
                        const int* grp_ptr, const int* seg_start, const int* seg_end, int max_rows, int causal, int64_t n_groups, int64_t Lq,
                        int64_t Lk, int64_t H, int64_t dh, float* dq, int64_t lddq, float* dk, float* dv, int64_t lddkv, hipStream_t s,
                        StltDrop dr, uint32_t site) {
  if (!q || !k || !v || !dctx || !dq || !dk || !dv) return stlt_set_error(STLT_EINVAL, "attention backward (head dim %lld): null pointer", (long long)dh);
  if (dh < 1 || dh > ANY_MAX_DH) return stlt_set_error(STLT_EINVAL, "attention backward: head dim %lld unsupported (1 ... %d)", (long long)dh, ANY_MAX_DH);
  const int64_t longest = grp_ptr ? max_rows : (Lq > Lk ? Lq : Lk);
  if (Lq <= 0 || Lk <= 0 || longest > ANY_MAX_SEQ)
    return stlt_set_error(STLT_EINVAL, "attention backward supports sequences of at most %d tokens (got %lld)", ANY_MAX_SEQ, (long long)longest);
  if (causal && !grp_ptr && Lq != Lk) return stlt_set_error(STLT_EINVAL, "attention backward: causal masking needs Lq == Lk");
  if (H <= 0 || H > 65535 || n_groups < 0 || n_groups * H > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "attention backward: too many sequences");
  if (n_groups == 0) return 0;
  AnyBwd a{q, k, v, dctx, ldq, ldkv, dq, dk, dv, lddq, lddkv, grp_ptr ? nullptr : kpm, grp_ptr, grp_ptr ? seg_start : nullptr, grp_ptr ? seg_end : nullptr,
           (int)Lq, (int)Lk, (int)H, (int)dh, causal, 1.0f / sqrtf((float)dh)};
  const dim3 grid((unsigned)(n_groups * H));
  if (rows_vectorise(dh, {ldq, ldkv}, {q, k, v, dctx}))
    hipLaunchKernelGGL((attn_any_bwd_kernel<true>), grid, dim3(64 * ANY_WAVES), 0, s, a, dr, site);
  else
    hipLaunchKernelGGL((attn_any_bwd_kernel<false>), grid, dim3(64 * ANY_WAVES), 0, s, a, dr, site);
  return stlt_check_launch("attn_any_bwd_kernel");
}
