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
// trip).  Q/K/V tiles are staged in LDS with fully coalesced 16-byte loads (a head row is 256 B).
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int LD = DH + 4;  // padded LDS row: conflict-free ds_read_b128 for the 16-lane groups
constexpr int TILE = 32;

__device__ __forceinline__ void stage_tile(float* __restrict__ dst, const float* __restrict__ src, int64_t ld,
                                           int rows_valid, int lane) {
  // 32 rows x 64 floats; float4 f = lane + 64*i -> row f/16, col4 f%16: 16 lanes cover one 256-B head row
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = lane + 64 * i;
    const int row = f >> 4, c4 = (f & 15) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < rows_valid) v = *reinterpret_cast<const f32x4*>(src + (int64_t)row * ld + c4);
    *reinterpret_cast<f32x4*>(dst + row * LD + c4) = v;
  }
}

__global__ __launch_bounds__(64) void attn_core_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ kpm,
                                                       int causal, int64_t n_tokens, int L, int H, int GL, int nt,
                                                       float scale, float* __restrict__ ctx) {
  __shared__ __attribute__((aligned(16))) float Qs[TILE * LD];
  __shared__ __attribute__((aligned(16))) float Ks[TILE * LD];
  __shared__ __attribute__((aligned(16))) float Vs[TILE * LD];
  __shared__ int kmeta[TILE];

  const int lane = threadIdx.x;
  const int li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y;
  const int64_t g = blockIdx.x / nt;
  const int qb = blockIdx.x % nt;
  const int64_t tok0 = g * GL;
  const int gvalid = (int)((n_tokens - tok0) < GL ? (n_tokens - tok0) : GL);  // tokens of this group
  const int d = H * DH;
  const int64_t ld = 3 * (int64_t)d;
  const int q_first = qb * TILE;
  const int q_rows = gvalid - q_first < TILE ? gvalid - q_first : TILE;
  if (q_rows <= 0) return;

  stage_tile(Qs, qkv + (tok0 + q_first) * ld + head * DH, ld, q_rows, lane);
  __syncthreads();
  f32x4 qf[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const f32x4*>(Qs + li * LD + 8 * c + 4 * lh);

  // this lane's query: position in the group, sequence id and position in the sequence
  const int qi = q_first + li;
  const int q_seq = qi / L, q_pos = qi - q_seq * L;

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const int kt_end = (causal && GL == L) ? qb + 1 : nt;  // causal: key tiles past the query tile are fully masked
  for (int kt = 0; kt < kt_end; ++kt) {
    const int k_first = kt * TILE;
    const int k_rows = gvalid - k_first < TILE ? gvalid - k_first : TILE;
    __syncthreads();  // previous tile's LDS reads are done
    stage_tile(Ks, qkv + (tok0 + k_first) * ld + d + head * DH, ld, k_rows, lane);
    stage_tile(Vs, qkv + (tok0 + k_first) * ld + 2 * d + head * DH, ld, k_rows, lane);
    if (lane < TILE) {
      const int kj = k_first + lane;
      int meta = -1;
      if (lane < k_rows && kpm[tok0 + kj] == 0) {
        const int ks = kj / L;
        meta = (ks << 16) | (kj - ks * L);
      }
      kmeta[lane] = meta;
    }
    __syncthreads();

    // Sᵀ[j][i] = sum_k K[j][k] Q[i][k]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + li * LD + 8 * c + 4 * lh);
#pragma unroll
      for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[c][e], st, 0, 0, 0);
    }

    // mask + online softmax; register r <-> key j = (r&3) + 8*(r>>2) + 4*lh
    float p[16];
    float m_tile = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int meta = kmeta[j];
      const bool ok = meta >= 0 && (meta >> 16) == q_seq && (!causal || (meta & 0xffff) <= q_pos);
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
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

    // Oᵀ[c][i] += sum_j V[j][c] P[j][i] ; MFMA step r sums keys j(r,0) and j(r,1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v0 = Vs[j * LD + li];
      const float v1 = Vs[j * LD + 32 + li];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, p[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, p[r], o1, 0, 0, 0);
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;  // fully masked row -> zeros
  if (li < q_rows) {
    float* orow = ctx + (tok0 + qi) * (int64_t)d + head * DH;
    // lane (i,h), register 4q..4q+3 <-> channels 8q + 4h + (0..3) (+32 for o1)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a = {o0[4 * q] * inv, o0[4 * q + 1] * inv, o0[4 * q + 2] * inv, o0[4 * q + 3] * inv};
      f32x4 b = {o1[4 * q] * inv, o1[4 * q + 1] * inv, o1[4 * q + 2] * inv, o1[4 * q + 3] * inv};
      *reinterpret_cast<f32x4*>(orow + 8 * q + 4 * lh) = a;
      *reinterpret_cast<f32x4*>(orow + 32 + 8 * q + 4 * lh) = b;
    }
  }
}

}  // namespace

int launch_attn(const float* qkv, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, int64_t dh,
                float* ctx, int kid, hipStream_t s) {
  if (!qkv || !kpm || !ctx) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: null pointer");
  if (dh != DH) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: head dim %lld unsupported (kernel is built for dh=64)", (long long)dh);
  if (L <= 0 || L > 32768 || H <= 0 || H > 65535) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: bad L=%lld H=%lld", (long long)L, (long long)H);
  if (S == 0) return 0;
  const int P = L <= 16 ? (int)(TILE / L) : 1;  // sequences packed per 32-token tile
  const int GL = P * (int)L;
  const int nt = (GL + TILE - 1) / TILE;
  const int64_t n_tokens = S * L;
  const int64_t groups = (S + P - 1) / P;
  if (groups * nt > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: too many tiles");
  StltProfScope ps(kid, s);
  dim3 grid((unsigned)(groups * nt), (unsigned)H);
  hipLaunchKernelGGL(attn_core_kernel, grid, dim3(64), 0, s, qkv, kpm, causal, n_tokens, (int)L, (int)H, GL, nt,
                     1.0f / sqrtf((float)dh), ctx);
  return stlt_check_launch("attn_core_kernel");
}
