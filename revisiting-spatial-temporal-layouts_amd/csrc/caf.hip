// CAF / CACNF forward on precomputed appearance features (reference src/modelling/models.py:230-271, 286-298, 328-549).
// Pure orchestration of the kernels the STLT path already has (MFMA linear, attention core in its cross-attention
// form, residual+LayerNorm, gather) plus three tiny data-movement kernels.
#include "common.h"

int backbone_impl_public(const stlt_params* p, const stlt_inputs* in, void* workspace, size_t workspace_bytes, int flags,
                         float* out_btd, hipStream_t s);
size_t stlt_workspace_bytes_public(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_classes);

namespace {

#define TRY(expr) do { int _e = (expr); if (_e) return _e; } while (0)
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// (B, C, S) feature map -> token-major (B*S, C)
__global__ __launch_bounds__(256) void feat_transpose_kernel(const float* __restrict__ f, int C, int S, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z;
  const int c0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, s = s0 + tx;
    tile[r][tx] = (c < C && s < S) ? f[(b * C + c) * (int64_t)S + s] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int s = s0 + r, c = c0 + tx;
    if (s < S && c < C) out[(b * S + s) * (int64_t)C + c] = tile[tx][r];
  }
}

// tokens (B, S+1, d): row 0 = cls + pos[0], row 1+s = proj[b,s] + pos[1+s]   (models.py:262-267)
__global__ __launch_bounds__(256) void app_assemble_kernel(const float* __restrict__ proj, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, int S, int d, float* __restrict__ out) {
  const int64_t row = blockIdx.x;  // b*(S+1) + t
  const int t = (int)(row % (S + 1));
  const int64_t b = row / (S + 1);
  for (int e = threadIdx.x * 4; e < d; e += 1024) {
    f32x4 v = t == 0 ? *reinterpret_cast<const f32x4*>(cls + e) : *reinterpret_cast<const f32x4*>(proj + (b * S + (t - 1)) * (int64_t)d + e);
    v += *reinterpret_cast<const f32x4*>(pos + (int64_t)t * d + e);
    *reinterpret_cast<f32x4*>(out + row * d + e) = v;
  }
}

// out[b] = [a[b*lda : +d], c[b*ldc : +d]]  (torch.cat(..., dim=-1), models.py:470-476)
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ c, int64_t ldc,
                                                      int d, float* __restrict__ out) {
  const int64_t b = blockIdx.x;
  for (int e = threadIdx.x * 4; e < 2 * d; e += 1024) {
    const f32x4 v = e < d ? *reinterpret_cast<const f32x4*>(a + b * lda + e) : *reinterpret_cast<const f32x4*>(c + b * ldc + (e - d));
    *reinterpret_cast<f32x4*>(out + b * 2 * (int64_t)d + e) = v;
  }
}

__global__ __launch_bounds__(256) void mean3_kernel(const float* a, const float* b, const float* c, float* o, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = ((a[i] + b[i]) + c[i]) / 3.0f;  // sum(logits) / 3, models.py:546
}

struct CafWs {
  size_t bb, lh, ah, la, aa, q, kv, ctx, tmp, hh, ft, proj, zero, fused, h1, h2, hl, sk, total;
};

CafWs caf_ws(int64_t B, int64_t T, int64_t N, int64_t d, int64_t C, int64_t S, int64_t K) {
  CafWs w;
  const size_t f = sizeof(float);
  const size_t rows = (size_t)B * (T > S + 1 ? T : S + 1);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
  w.bb = take(stlt_workspace_bytes_public(B, T, N, d, K));
  w.lh = take((size_t)B * T * d * f);
  w.ah = take((size_t)B * (S + 1) * d * f);
  w.la = take(rows * d * f);
  w.aa = take(rows * d * f);
  w.q = take(rows * 3 * d * f);   // also holds a packed qkv for the self-attention layers
  w.kv = take(rows * 2 * d * f);
  w.ctx = take(rows * d * f);
  w.tmp = take(rows * d * f);
  w.hh = take(rows * 4 * d * f);
  w.ft = take((size_t)B * S * C * f);
  w.proj = take((size_t)B * S * d * f);
  w.zero = take(rows);            // all-zero key-padding mask
  w.fused = take((size_t)B * 2 * d * f);
  w.h1 = take((size_t)B * d * f);
  w.h2 = take((size_t)B * d * f);
  w.hl = take((size_t)B * d * f);
  w.sk = take(STLT_GEMM_SCRATCH_BYTES);
  w.total = off;
  return w;
}

struct Bufs { float *q, *kv, *ctx, *tmp, *hh; const uint8_t* zero; };

// SelfAttentionLayer (models.py:345-360): out = LN(MHA(x,x,x) + x)
int self_attn_block(const stlt_attn_block_params& p, int64_t d, int64_t H, float eps, const float* x, int64_t S, int64_t L,
                    const uint8_t* kpm, int causal, const Bufs& b, float* out, hipStream_t s) {
  const int64_t M = S * L;
  TRY(launch_linear(x, d, p.in_proj_w, p.in_proj_b, b.q, 3 * d, M, 3 * d, d, STLT_ACT_NONE, s));
  TRY(launch_attn(b.q, kpm ? kpm : b.zero, causal, S, L, H, d / H, b.ctx, causal ? STLT_K_ATTN_TEMPORAL : STLT_K_ATTN_SPATIAL, s));
  TRY(launch_linear(b.ctx, d, p.out_proj_w, p.out_proj_b, b.tmp, d, M, d, d, STLT_ACT_NONE, s));
  return launch_add_layernorm(b.tmp, d, x, d, p.ln_w, p.ln_b, eps, M, d, out, d, s);
}

// CrossAttentionLayer (models.py:362-382): out = LN(MHA(x, ctx, ctx, key_padding_mask) + x)
int cross_attn_block(const stlt_attn_block_params& p, int64_t d, int64_t H, float eps, const float* x, int64_t Lq, const float* c,
                     int64_t Lk, const uint8_t* kpm_k, int64_t S, const Bufs& b, float* out, hipStream_t s) {
  TRY(launch_linear(x, d, p.in_proj_w, p.in_proj_b, b.q, d, S * Lq, d, d, STLT_ACT_NONE, s));                      // q rows of in_proj
  TRY(launch_linear(c, d, p.in_proj_w + d * d, p.in_proj_b + d, b.kv, 2 * d, S * Lk, 2 * d, d, STLT_ACT_NONE, s));  // k,v rows
  TRY(launch_attn_general(b.q, d, b.kv, b.kv + d, 2 * d, kpm_k ? kpm_k : b.zero, 0, S, Lq, Lk, H, d / H, b.ctx, STLT_K_ATTN_SPATIAL, s));
  TRY(launch_linear(b.ctx, d, p.out_proj_w, p.out_proj_b, b.tmp, d, S * Lq, d, d, STLT_ACT_NONE, s));
  return launch_add_layernorm(b.tmp, d, x, d, p.ln_w, p.ln_b, eps, S * Lq, d, out, d, s);
}

int head_block(const stlt_head_params& h, const float* x, int64_t ldx, int64_t in_dim, int64_t B, int64_t d, int64_t K, float eps,
               float* h1, float* h2, float* logits, hipStream_t s) {
  TRY(launch_linear(x, ldx, h.fc1_w, h.fc1_b, h1, d, B, d, in_dim, STLT_ACT_GELU, s));
  TRY(launch_add_layernorm(h1, d, nullptr, 0, h.ln_w, h.ln_b, eps, B, d, h2, d, s));
  return launch_linear(h2, d, h.fc2_w, h.fc2_b, logits, K, B, K, d, STLT_ACT_NONE, s);
}

}  // namespace

extern "C" size_t stlt_caf_workspace_bytes(int64_t B, int64_t T, int64_t N, int64_t d, int64_t feat_channels, int64_t app_tokens,
                                           int64_t n_classes) {
  if (B <= 0 || T <= 0 || N <= 0 || d <= 0 || feat_channels <= 0 || app_tokens <= 0) return 0;
  return caf_ws(B, T, N, d, feat_channels, app_tokens, n_classes).total;
}

extern "C" int stlt_caf_forward(const stlt_caf_params* p, const stlt_inputs* in, const float* feats, void* workspace,
                                size_t workspace_bytes, float* logits_caf, float* logits_stlt, float* logits_resnet3d,
                                float* logits_ensemble, stlt_stream_t stream) {
  return stlt_caf_forward_flags(p, in, feats, workspace, workspace_bytes, 0, logits_caf, logits_stlt, logits_resnet3d, logits_ensemble, stream);
}

extern "C" int stlt_caf_forward_flags(const stlt_caf_params* p, const stlt_inputs* in, const float* feats, void* workspace,
                                      size_t workspace_bytes, int flags, float* logits_caf, float* logits_stlt,
                                      float* logits_resnet3d, float* logits_ensemble, stlt_stream_t stream) {
  if (!p || !in || !feats || !workspace || !logits_caf) return stlt_set_error(STLT_EINVAL, "stlt_caf_forward: null argument");
  hipStream_t s = (hipStream_t)stream;
  const stlt_params& lp = p->layout;
  const int64_t B = in->B, T = in->T, N = in->N, d = lp.d, H = lp.H, C = p->feat_channels, S = p->app_tokens, K = lp.n_classes;
  const float eps = lp.ln_eps;
  if (K <= 0 || !in->lengths || !p->fusion_head.fc1_w) return stlt_set_error(STLT_EINVAL, "stlt_caf_forward: heads / lengths missing");
  if (B <= 0 || T <= 0 || N <= 0 || C <= 0 || S <= 0 || T > 256 || S > 4096 || N > 4096 || B > (int64_t)0x7fffffff / (T * N) || B > (int64_t)0x7fffffff / (S + 1) || K > 65536 ||
      C > 65536 || p->n_app_layers < 0 || p->n_fusion < 0 || (p->n_app_layers > 0 && !p->app_layers) || (p->n_fusion > 0 && !p->fusion))
    return stlt_set_error(STLT_EINVAL, "stlt_caf_forward: bad shape (B %lld, T %lld, N %lld, feature channels %lld, appearance tokens %lld, classes %lld)", (long long)B,
                          (long long)T, (long long)N, (long long)C, (long long)S, (long long)K);
  if (C % 4 != 0 || !stlt_heads_ok(d, H)) return stlt_set_error(STLT_EINVAL, "stlt_caf_forward: feat_channels and hidden_size must be multiples of 4, hidden_size %% heads == 0, head dim <= 256");
  const bool cacnf = p->layout_head.fc1_w != nullptr;
  if (cacnf && (!p->appearance_head.fc1_w || !logits_stlt || !logits_resnet3d || !logits_ensemble))
    return stlt_set_error(STLT_EINVAL, "stlt_caf_forward: CACNF needs both unimodal heads and all four outputs");
  const CafWs w = caf_ws(B, T, N, d, C, S, K);
  if (workspace_bytes < w.total) return stlt_set_error(STLT_EWORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.total);
  char* base = (char*)workspace;
  StltGemmScratch gemm_scratch(base + w.sk, STLT_GEMM_SCRATCH_BYTES);
  auto F = [&](size_t o) { return (float*)(base + o); };
  float *Lh = F(w.lh), *Ah = F(w.ah), *la = F(w.la), *aa = F(w.aa);
  Bufs b{F(w.q), F(w.kv), F(w.ctx), F(w.tmp), F(w.hh), (const uint8_t*)(base + w.zero)};
  if (hipError_t e = hipMemsetAsync(base + w.zero, 0, (size_t)B * (T > S + 1 ? T : S + 1), s); e != hipSuccess)
    return stlt_set_error((int)e, "stlt_caf_forward: memset: %s", hipGetErrorString(e));

  // ---- layout branch: full (B,T,d) backbone output (models.py:451); STLT_FLAG_SKIP_PADDING computes it on the real
  // tokens / frames only and leaves the padded frames' rows zero (they are masked keys everywhere downstream)
  TRY(backbone_impl_public(&lp, in, base + w.bb, stlt_workspace_bytes_public(B, T, N, d, K),
                           STLT_FLAG_CLS_ONLY_LAST_SPATIAL | (flags & STLT_FLAG_SKIP_PADDING), Lh, s));
  // ---- appearance branch from the feature map (models.py:253-271)
  hipLaunchKernelGGL(feat_transpose_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((S + 31) / 32), (unsigned)B), dim3(256), 0, s, feats,
                     (int)C, (int)S, F(w.ft));
  TRY(stlt_check_launch("feat_transpose_kernel"));
  TRY(launch_linear(F(w.ft), C, p->proj_w, p->proj_b, F(w.proj), d, B * S, d, C, STLT_ACT_NONE, s));
  hipLaunchKernelGGL(app_assemble_kernel, dim3((unsigned)(B * (S + 1))), dim3(256), 0, s, F(w.proj), p->cls_token, p->pos_embed, (int)S,
                     (int)d, Ah);
  TRY(stlt_check_launch("app_assemble_kernel"));
  const int64_t LA = S + 1, MA = B * LA, ML = B * T;
  for (int64_t l = 0; l < p->n_app_layers; ++l) {  // ReLU post-norm encoder layers, eps 1e-5, no masks
    const stlt_layer_params& e = p->app_layers[l];
    TRY(launch_linear(Ah, d, e.in_proj_w, e.in_proj_b, b.q, 3 * d, MA, 3 * d, d, STLT_ACT_NONE, s));
    TRY(launch_attn(b.q, b.zero, 0, B, LA, H, d / H, b.ctx, STLT_K_ATTN_SPATIAL, s));
    TRY(launch_linear(b.ctx, d, e.out_proj_w, e.out_proj_b, b.tmp, d, MA, d, d, STLT_ACT_NONE, s));
    TRY(launch_add_layernorm(b.tmp, d, Ah, d, e.norm1_w, e.norm1_b, 1e-5f, MA, d, aa, d, s));
    TRY(launch_linear(aa, d, e.lin1_w, e.lin1_b, b.hh, 4 * d, MA, 4 * d, d, STLT_ACT_RELU, s));
    TRY(launch_linear(b.hh, 4 * d, e.lin2_w, e.lin2_b, b.tmp, d, MA, d, 4 * d, STLT_ACT_NONE, s));
    TRY(launch_add_layernorm(b.tmp, d, aa, d, e.norm2_w, e.norm2_b, 1e-5f, MA, d, Ah, d, s));
  }
  // ---- unimodal states before fusion (models.py:459-460) -> CACNF heads
  if (cacnf) {
    TRY(launch_gather_last(Lh, in->lengths, B, T, d, F(w.hl), s));
    TRY(head_block(p->layout_head, F(w.hl), d, d, B, d, K, eps, F(w.h1), F(w.h2), logits_stlt, s));
    TRY(head_block(p->appearance_head, Ah, LA * d, d, B, d, K, eps, F(w.h1), F(w.h2), logits_resnet3d, s));  // rows (b, token 0)
  }
  // ---- multimodal fusion (models.py:462-468, 403-431)
  for (int64_t l = 0; l < p->n_fusion; ++l) {
    const stlt_crossmodal_params& m = p->fusion[l];
    TRY(cross_attn_block(m.cross_attn, d, H, eps, Lh, T, Ah, LA, nullptr, B, b, la, s));             // layout <- appearance
    TRY(cross_attn_block(m.cross_attn, d, H, eps, Ah, LA, Lh, T, in->kpm_frames, B, b, aa, s));      // appearance <- layout
    TRY(self_attn_block(m.layout_attn, d, H, eps, la, B, T, in->kpm_frames, 1, b, Lh, s));           // Lh = layout self-attn
    TRY(self_attn_block(m.appearance_attn, d, H, eps, aa, B, LA, nullptr, 0, b, Ah, s));             // Ah = appearance self-attn
    // layout_ffn: LN(lin2(gelu(lin1(x))) + x)
    TRY(launch_linear(Lh, d, m.layout_ffn.lin1_w, m.layout_ffn.lin1_b, b.hh, 4 * d, ML, 4 * d, d, STLT_ACT_GELU, s));
    TRY(launch_linear(b.hh, 4 * d, m.layout_ffn.lin2_w, m.layout_ffn.lin2_b, b.tmp, d, ML, d, 4 * d, STLT_ACT_NONE, s));
    TRY(launch_add_layernorm(b.tmp, d, Lh, d, m.layout_ffn.ln_w, m.layout_ffn.ln_b, eps, ML, d, la, d, s));
    // appearance_ffn is a SelfAttentionLayer in the reference (models.py:401)
    TRY(self_attn_block(m.appearance_ffn, d, H, eps, Ah, B, LA, nullptr, 0, b, aa, s));
    float* t1 = Lh; Lh = la; la = t1;   // outputs of this module feed the next one
    float* t2 = Ah; Ah = aa; aa = t2;
  }
  // ---- fused state + FusionHead (models.py:470-476, 286-298)
  TRY(launch_gather_last(Lh, in->lengths, B, T, d, F(w.hl), s));
  hipLaunchKernelGGL(concat2_kernel, dim3((unsigned)B), dim3(256), 0, s, F(w.hl), d, Ah, LA * d, (int)d, F(w.fused));
  TRY(stlt_check_launch("concat2_kernel"));
  TRY(head_block(p->fusion_head, F(w.fused), 2 * d, 2 * d, B, d, K, eps, F(w.h1), F(w.h2), logits_caf, s));
  if (cacnf) {
    hipLaunchKernelGGL(mean3_kernel, dim3((unsigned)((B * K + 255) / 256)), dim3(256), 0, s, logits_stlt, logits_resnet3d, logits_caf,
                       logits_ensemble, B * K);
    TRY(stlt_check_launch("mean3_kernel"));
  }
  return 0;
}
