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

__global__ __launch_bounds__(64) void attn_core_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ kpm,
                                                       int causal, int64_t n_tokens, int L, int H, int GL, int nt,
                                                       float scale, float* __restrict__ ctx) {
  __shared__ __attribute__((aligned(16))) float smem[3 * TILE_FLOATS + TILE];
  float* Qs = smem;
  float* Ks = smem + TILE_FLOATS;
  float* Vs = smem + 2 * TILE_FLOATS;
  int* kmeta = reinterpret_cast<int*>(smem + 3 * TILE_FLOATS);

  const int lane = threadIdx.x;
  const int li = lane & 31, lh = lane >> 5;
  // head-fastest block order: the waves resident at any moment cover all heads of the same tokens, i.e. whole
  // contiguous qkv rows.
  const int head = blockIdx.x % H;
  const int64_t tile_id = blockIdx.x / H;
  const int64_t g = tile_id / nt;
  const int qb = (int)(tile_id % nt);
  const int64_t tok0 = g * GL;
  const int gvalid = (int)((n_tokens - tok0) < GL ? (n_tokens - tok0) : GL);  // tokens of this group
  const int d = H * DH;
  const int64_t ld = 3 * (int64_t)d;
  const int q_first = qb * TILE;
  const int q_rows = gvalid - q_first < TILE ? gvalid - q_first : TILE;
  if (q_rows <= 0) return;

  // this lane's query: position in the group, sequence id and position in the sequence
  const int qi = q_first + li;
  const int q_seq = qi / L, q_pos = qi - q_seq * L;

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  f32x4 qf[8];

  const int kt_end = (causal && GL == L) ? qb + 1 : nt;  // causal: key tiles past the query tile are fully masked
  for (int kt = 0; kt < kt_end; ++kt) {
    const int k_first = kt * TILE;
    const int k_rows = gvalid - k_first < TILE ? gvalid - k_first : TILE;
    if (kt > 0) __syncthreads();  // previous tile's LDS reads are done
    // every byte of the tile (and, first time round, the queries) is in flight before the first wait
    if (kt == 0) tile_dma(Qs, qkv + (tok0 + q_first) * ld + head * DH, ld, q_rows, lane);
    tile_dma(Ks, qkv + (tok0 + k_first) * ld + d + head * DH, ld, k_rows, lane);
    tile_dma(Vs, qkv + (tok0 + k_first) * ld + 2 * d + head * DH, ld, k_rows, lane);
    {
      // key metadata for the mask: -1 = masked/absent, else (sequence id << 16) | position in sequence
      const int kj = k_first + li;
      const bool kvalid = li < k_rows;
      const uint8_t pad = kpm[tok0 + (kvalid ? kj : k_first)];
      const int ks = kj / L;
      if (lane < TILE) kmeta[lane] = (kvalid && pad == 0) ? ((ks << 16) | (kj - ks * L)) : -1;
    }
    __syncthreads();  // waits vmcnt(0) (LDS-DMA landed) + lgkmcnt(0)
    if (kt == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const f32x4*>(Qs + li * DH + swz(li, 2 * c + lh) * 4);
    }

    // Sᵀ[j][i] = sum_k K[j][k] Q[i][k]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + li * DH + swz(li, 2 * c + lh) * 4);
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
    if (kt > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }

    // Oᵀ[c][i] += sum_j V[j][c] P[j][i] ; MFMA step r sums keys j(r,0) and j(r,1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float* vrow = Vs + j * DH + (li & 3);
      const float v0 = vrow[swz(j, li >> 2) * 4];
      const float v1 = vrow[swz(j, 8 + (li >> 2)) * 4];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, p[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, p[r], o1, 0, 0, 0);
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;  // fully masked row -> zeros
  // Epilogue through LDS (the Q tile is dead: its fragments live in registers) so that every global store
  // instruction writes whole 256-B head rows.  lane (i,h), registers 4q..4q+3 <-> channels 8q+4h+(0..3) (+32 for o1)
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 a = {o0[4 * q] * inv, o0[4 * q + 1] * inv, o0[4 * q + 2] * inv, o0[4 * q + 3] * inv};
    f32x4 b = {o1[4 * q] * inv, o1[4 * q + 1] * inv, o1[4 * q + 2] * inv, o1[4 * q + 3] * inv};
    *reinterpret_cast<f32x4*>(Qs + li * DH + swz(li, 2 * q + lh) * 4) = a;
    *reinterpret_cast<f32x4*>(Qs + li * DH + swz(li, 8 + 2 * q + lh) * 4) = b;
  }
  __syncthreads();
  float* obase = ctx + (tok0 + q_first) * (int64_t)d + head * DH;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + (lane >> 4), chunk = lane & 15;
    const f32x4 v = *reinterpret_cast<const f32x4*>(Qs + row * DH + swz(row, chunk) * 4);
    if (row < q_rows) *reinterpret_cast<f32x4*>(obase + (int64_t)row * d + chunk * 4) = v;
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
  if (groups * nt * H > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "stlt_attn_core_fwd: too many tiles");
  StltProfScope ps(kid, s);
  dim3 grid((unsigned)(groups * nt * H));
  hipLaunchKernelGGL(attn_core_kernel, grid, dim3(64), 0, s, qkv, kpm, causal, n_tokens, (int)L, (int)H, GL, nt,
                     1.0f / sqrtf((float)dh), ctx);
  return stlt_check_launch("attn_core_kernel");
}
