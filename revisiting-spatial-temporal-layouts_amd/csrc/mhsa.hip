// Fused in-projection + causal attention core for the temporal tower (SURVEY.md §8 row N1 / the north star's "fused MHSA"):
//   ctx[b, :, h] = softmax(q_h k_hᵀ / 8 + causal + key padding) v_h   with   [q_h | k_h | v_h] = X_b · W_in[h]ᵀ + b_in[h]
// for clips of exactly 32 frames and 64-channel heads (nn.MultiheadAttention as configured at models.py:118-124; mask of
// utils/model_utils.py:4-7).  The packed QKV tensor never goes to HBM: only X is read and ctx written.
//
// Work item = (group of 4 clips = 128 token rows, head): a 128 x 192 x d product on the f32 MFMA (the 192 output columns
// are the head's 64 q, k and v channels), then the four 32 x 32 attention problems of the group, all in one workgroup:
//   * 8 MFMA waves as (clip mb = wave & 3, channel half nh = wave >> 2): a wave owns one clip's 32 rows and channels
//     [32 nh, 32 nh + 32) of q, k and v — three 32x32 MFMA tiles (48 accumulators), computed transposed (D[channel][token])
//     like gemm.hip, so a lane owns a token and its registers are channels;
//   * in that layout the q and k accumulators ARE the MFMA operands of Sᵀ = K Qᵀ over the wave's 32 channels (register r
//     of both operands is the same channel pair), no data movement; the two channel halves of a clip add their partial
//     scores through LDS (4 KB each way);
//   * mask + softmax over a lane's 32 keys in registers (one cross-half shuffle), probabilities stay in registers as the B
//     operand of Oᵀ = Vᵀ Pᵀ; the wave's v half goes through a private 32x33 LDS tile (the A operand wants lane = channel);
//   * each lane stores its token's 32 output channels with four 16-byte stores.
// Operand staging is gemm.hip's: LDS-DMA with the source-side bank swizzle, three 40-KB stages, four DMA-only loader waves
// running two k-steps ahead with a counted vmcnt, one barrier per k-step, the bias strip DMA'd one item ahead; the loaders
// keep the next item's first k-steps in flight while the MFMA waves are in the attention phase.
#include "common.h"

namespace {

constexpr int FM = 128, FN = 192, FK = 32;
constexpr int F_WAVES = 8, F_LOADERS = 4;
constexpr int F_THREADS = 64 * (F_WAVES + F_LOADERS);
constexpr int F_NSTAGE = 3;
constexpr int F_STAGE = (FM + FN) * FK;        // 10240 floats = 40 KB
constexpr int F_ATT_WAVE = 1088;               // per-wave attention scratch: 16x64 partial scores, then the 32x33 v tile
constexpr int F_SMEM = F_NSTAGE * F_STAGE + F_WAVES * F_ATT_WAVE + 2 * FN;  // 39808 floats = 159232 B

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

__global__ __launch_bounds__(F_THREADS, 3) void mhsa_fused_kernel(const float* __restrict__ X, const float* __restrict__ Win,
                                                                  const float* __restrict__ bin, const uint8_t* __restrict__ kpm,
                                                                  float* __restrict__ ctx, int n_clips, int H, int d, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nk = d / FK;
  const int M = n_clips * 32;
  const int n_groups = (n_clips + 3) >> 2;
  const int n_items = n_groups * H;  // head fastest: the 12 heads of a clip group sit on neighbouring workgroups of one XCD
  const int G = gridDim.x;
  int v = blockIdx.x;
  if ((G & 7) == 0) v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);  // XCD-contiguous virtual id (round-robin dispatch)
  const int my_items = (n_items - v + G - 1) / G;
  if (my_items <= 0) return;
  const int total_steps = my_items * nk;
  float* bias_lds = smem + F_NSTAGE * F_STAGE + F_WAVES * F_ATT_WAVE;
  auto item_of = [&](int it, int& grp, int& head) {
    const int item = v + it * G;
    grp = item / H;
    head = item - grp * H;
  };

  if (wave >= F_WAVES) {
    // ---- loader waves: the whole DMA stream of the workgroup.  Loader L issues A rows [32L, 32L+32) and B rows [48L, 48L+48).
    const int L = wave - F_WAVES;
    const int drow = lane >> 3, dslot = lane & 7;
    const float* pa[4];
    const float* pb[6];
    auto set_item = [&](int it) {
      int grp, head;
      item_of(it, grp, head);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = L * 32 + i * 8 + drow;
        int gm = grp * FM + r;
        gm = gm < M ? gm : M - 1;  // a ragged last group re-reads the last row; its stores are guarded
        pa[i] = X + (int64_t)gm * d + (dslot ^ ((r >> 1) & 7)) * 4;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int r = L * 48 + i * 8 + drow;           // image row: (channel half, q|k|v, channel) = (r / 96, (r % 96) / 32, r % 32)
        const int half = r / 96, rem = r - half * 96, which = rem >> 5, ch = rem & 31;
        const int wrow = which * d + head * 64 + half * 32 + ch;
        pb[i] = Win + (int64_t)wrow * d + (dslot ^ ((r >> 1) & 7)) * 4;
      }
    };
    auto dma_bias = [&](int it) {
      if (L == 0) {
        int grp, head;
        item_of(it, grp, head);
        float* dst = bias_lds + (it & 1) * FN;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int r = i * 64 + lane;
          const int half = r / 96, rem = r - half * 96, which = rem >> 5, ch = rem & 31;
          __builtin_amdgcn_global_load_lds((glb_void_ptr)(bin + which * d + head * 64 + half * 32 + ch), (lds_void_ptr)(dst + i * 64), 4, 0, 0);
        }
      }
    };
    int l_it = 0, l_kt = 0, l_stage = 0;
    auto l_step = [&]() {
      if (l_kt == 0) set_item(l_it);
      float* sa = smem + l_stage * F_STAGE + (L * 32) * FK;
      float* sb = smem + l_stage * F_STAGE + FM * FK + (L * 48) * FK;
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_void_ptr)(pa[i] + l_kt * FK), (lds_void_ptr)(sa + i * 8 * FK), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) __builtin_amdgcn_global_load_lds((glb_void_ptr)(pb[i] + l_kt * FK), (lds_void_ptr)(sb + i * 8 * FK), 16, 0, 0);
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
        __builtin_amdgcn_s_barrier();  // the two barriers of the MFMA waves' attention phase
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }

  // ---- MFMA waves
  const int mb = wave & 3, nh = wave >> 2;
  const int sw = (lr >> 1) & 7;
  const int a_row = (mb * 32 + lr) * FK;
  const int b_row = (FM + nh * 96 + lr) * FK;  // + 32*FK per tile (q, k, v)
  float* att = smem + F_NSTAGE * F_STAGE + wave * F_ATT_WAVE;
  float* att_partner = smem + F_NSTAGE * F_STAGE + (wave ^ 4) * F_ATT_WAVE;
  struct Frags { f32x4 a, b0, b1, b2; };
  auto read_frags = [&](int stage, int c) {
    const float* s = smem + stage * F_STAGE;
    const int off = ((2 * c + lh) ^ sw) * 4;
    Frags f;
    f.a = *reinterpret_cast<const f32x4*>(s + a_row + off);
    f.b0 = *reinterpret_cast<const f32x4*>(s + b_row + off);
    f.b1 = *reinterpret_cast<const f32x4*>(s + b_row + 32 * FK + off);
    f.b2 = *reinterpret_cast<const f32x4*>(s + b_row + 64 * FK + off);
    return f;
  };
  f32x16 aq, ak, av;  // D[channel][token]: register r of lane (lr, lh) = channel (r&3) + 8*(r>>2) + 4*lh of token lr
  auto init_acc = [&](int it) {
    const float* src = bias_lds + (it & 1) * FN + nh * 96 + 4 * lh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 vq = *reinterpret_cast<const f32x4*>(src + 8 * q);
      const f32x4 vk = *reinterpret_cast<const f32x4*>(src + 32 + 8 * q);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(src + 64 + 8 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) { aq[4 * q + j] = vq[j]; ak[4 * q + j] = vk[j]; av[4 * q + j] = vv[j]; }
    }
  };
  auto mfma_chunk = [&](const Frags& f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      aq = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b0[e], f.a[e], aq, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b1[e], f.a[e], ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b2[e], f.a[e], av, 0, 0, 0);
    }
  };

  __builtin_amdgcn_s_barrier();  // the loaders' counted wait + this barrier publish step 0 and the first bias strip
  init_acc(0);
  int c_it = 0, c_kt = 0, stage = 0;
  Frags fa = read_frags(0, 0), fb;
  for (int step = 0; step < total_steps; ++step) {
    const int next_stage = stage + 1 == F_NSTAGE ? 0 : stage + 1;
    fb = read_frags(stage, 1);
    mfma_chunk(fa);
    fa = read_frags(stage, 2);
    mfma_chunk(fb);
    fb = read_frags(stage, 3);
    mfma_chunk(fa);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own fragment reads of this stage are done
    __builtin_amdgcn_s_barrier();                        // retire the stage; step+1 landed
    if (step + 1 < total_steps) fa = read_frags(next_stage, 0);
    mfma_chunk(fb);
    stage = next_stage;
    if (++c_kt < nk) continue;

    // ---- attention phase of item c_it: this wave holds q, k, v channels [32 nh, 32 nh + 32) of clip mb
    int grp, head;
    item_of(c_it, grp, head);
    const int clip = grp * 4 + mb;
    const bool clip_ok = clip < n_clips;
    // partial scores over the wave's 32 channels: Sᵀ[key][query] (lane = query, register r = key (r&3) + 8*(r>>2) + 4*lh)
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) st = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[r], aq[r], st, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) att[r * 64 + lane] = st[r];
    // key padding of this clip: bit j of `kmask` = key j is padded (each of the 32 low lanes looks at its own frame)
    const unsigned pad_byte = clip_ok ? (unsigned)kpm[clip * 32 + lr] : 1u;
    const unsigned kmask = (unsigned)__ballot(pad_byte != 0);  // lanes 0..31 = frames 0..31 (lanes 32..63 repeat them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // attention barrier 1: the partner's partial scores are in LDS
    float p[16];
    float m_row = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float other = att_partner[r * 64 + lane];
      const float s_full = nh == 0 ? st[r] + other : other + st[r];  // low channel half first in both waves: identical sums
      const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const bool ok = (j <= lr) & (((kmask >> j) & 1u) == 0u);
      p[r] = ok ? s_full * scale : -1e30f;
      m_row = fmaxf(m_row, p[r]);
    }
    m_row = fmaxf(m_row, __shfl_xor(m_row, 32, 64));
    float l_row = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = p[r] > -1e29f ? __expf(p[r] - m_row) : 0.f;
      l_row += p[r];
    }
    l_row += __shfl_xor(l_row, 32, 64);
    const float inv = l_row > 0.f ? 1.0f / l_row : 0.f;  // fully masked row -> zeros
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // attention barrier 2: both waves of the pair have read the partial scores
    // v half -> private LDS tile vs[key][channel] (row stride 33): the A operand of Oᵀ = Vᵀ Pᵀ wants lane = channel
#pragma unroll
    for (int r = 0; r < 16; ++r) att[lr * 33 + (r & 3) + 8 * (r >> 2) + 4 * lh] = av[r];
    if (c_it + 1 < my_items) init_acc(c_it + 1);  // q, k, v are consumed: the next item's bias strip is published (k-step barrier)
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // DS operations of a wave complete in order: the tile is written
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(att[j * 33 + lr], p[r], o, 0, 0, 0);
    }
    if (clip_ok) {
      float* orow = ctx + (int64_t)(clip * 32 + lr) * d + head * 64 + nh * 32 + 4 * lh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 val = {o[4 * q] * inv, o[4 * q + 1] * inv, o[4 * q + 2] * inv, o[4 * q + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 8 * q) = val;
      }
    }
    ++c_it;
    c_kt = 0;
  }
}

}  // namespace

// ctx (n_clips*32, d) = causal multi-head attention of every 32-frame clip with the in-projection fused in.
// x (n_clips*32, d), w_in (3d, d) rows [q; k; v], b_in (3d), kpm (n_clips*32) bytes (1 = padded frame).
int launch_mhsa_fused(const float* x, const float* w_in, const float* b_in, const uint8_t* kpm, int64_t n_clips, int64_t T, int64_t H,
                      int64_t d, float* ctx, hipStream_t s) {
  if (!x || !w_in || !b_in || !kpm || !ctx) return stlt_set_error(STLT_EINVAL, "mhsa_fused: null pointer");
  if (T != 32 || H <= 0 || d != H * 64 || d % FK != 0)
    return stlt_set_error(STLT_EINVAL, "mhsa_fused: clips of exactly 32 frames and 64-channel heads (T=%lld, d=%lld, H=%lld)", (long long)T, (long long)d, (long long)H);
  if (n_clips <= 0) return 0;
  if (n_clips * 32 * d > 0x7fffffffLL * 4) return stlt_set_error(STLT_EINVAL, "mhsa_fused: batch too large");
  static StltPerDeviceOnce attr_done;
  if (!attr_done.flag()) {
    if (hipError_t e = hipFuncSetAttribute((const void*)mhsa_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_SMEM * (int)sizeof(float)); e != hipSuccess)
      return stlt_set_error((int)e, "mhsa_fused: %s", hipGetErrorString(e));
    attr_done.flag() = true;
  }
  const int64_t n_items = (n_clips + 3) / 4 * H;
  int64_t G = stlt_device_cus();
  if (G > n_items) G = n_items;
  StltProfScope ps(STLT_K_MHSA_FUSED, s);
  hipLaunchKernelGGL(mhsa_fused_kernel, dim3((unsigned)G), dim3(F_THREADS), F_SMEM * sizeof(float), s, x, w_in, b_in, kpm, ctx, (int)n_clips, (int)H,
                     (int)d, 0.125f);
  return stlt_check_launch("mhsa_fused_kernel");
}
