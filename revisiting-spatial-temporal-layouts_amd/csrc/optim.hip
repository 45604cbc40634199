// Gradient clipping + AdamW of the training step (reference train.py:128-131: clip_grad_norm_(5.0), AdamW.step) as
// two passes over ONE flat gradient buffer: the reverse sweep already writes every parameter gradient into one
// contiguous allocation (the buffer a data-parallel run all-reduces), so the norm is one reduction and the update is
// one kernel over a table of (parameter pointer, flat offset, length, weight decay) chunks, instead of torch's
// ~10 multi-tensor launches.  Arithmetic follows torch.optim.AdamW's foreach path operation by operation (fp32).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n4, int64_t n, float* __restrict__ partials) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 * 4; i < n; ++i) acc += g[i] * g[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// out[0] = ||g||_2, out[1] = min(1, max_norm / (||g|| + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void norm_finish_kernel(const float* __restrict__ partials, int n_partials, float max_norm,
                                                          float* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += 256) acc += (double)partials[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    out[0] = norm;
    out[1] = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(const stlt_opt_chunk* __restrict__ chunks, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, const float* __restrict__ clip,
                                                    float lr, float w_lerp, float beta2, float w_sq, float eps, float neg_step,
                                                    float bc2_sqrt) {
  const stlt_opt_chunk c = chunks[blockIdx.x];
  const float coef = clip ? clip[1] : 1.0f;
  const float decay = 1.0f - lr * c.weight_decay;  // param.mul_(1 - lr * weight_decay)
  float* __restrict__ p = c.param;
  auto update = [&](float gi, float& mi, float& vi, float& pi) {
    gi *= coef;
    pi = pi * decay;
    mi = mi + w_lerp * (gi - mi);              // exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * beta2;                           // exp_avg_sq.mul_(beta2)
    vi = vi + (w_sq * gi) * gi;                //            .addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi + neg_step * (mi / denom);         // param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
  };
  const bool vec = (((uintptr_t)p | (uintptr_t)(g + c.flat_offset) | (uintptr_t)(m + c.flat_offset) | (uintptr_t)(v + c.flat_offset)) & 15) == 0;
  const int n4 = vec ? c.n / 4 : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const int64_t k = c.flat_offset + 4 * (int64_t)i;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + k);
    f32x4 m4 = *reinterpret_cast<const f32x4*>(m + k), v4 = *reinterpret_cast<const f32x4*>(v + k);
    f32x4 p4 = *reinterpret_cast<const f32x4*>(p + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mi = m4[e], vi = v4[e], pi = p4[e];
      update(g4[e], mi, vi, pi);
      m4[e] = mi; v4[e] = vi; p4[e] = pi;
    }
    *reinterpret_cast<f32x4*>(m + k) = m4;
    *reinterpret_cast<f32x4*>(v + k) = v4;
    *reinterpret_cast<f32x4*>(p + 4 * i) = p4;
  }
  for (int i = n4 * 4 + threadIdx.x; i < c.n; i += 256) {
    const int64_t k = c.flat_offset + i;
    float mi = m[k], vi = v[k], pi = p[i];
    update(g[k], mi, vi, pi);
    m[k] = mi; v[k] = vi; p[i] = pi;
  }
}

}  // namespace

extern "C" {

int stlt_grad_norm(const float* flat_grad, int64_t n, float max_norm, float* scratch, float* out, stlt_stream_t stream) {
  if (!flat_grad || !scratch || !out) return stlt_set_error(STLT_EINVAL, "stlt_grad_norm: null pointer");
  if (n < 0 || ((uintptr_t)flat_grad & 15)) return stlt_set_error(STLT_EINVAL, "stlt_grad_norm: buffer must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  StltProfScope ps(STLT_K_OPTIM, s);
  stlt_prof_note("grad_norm n=%lld", (long long)n);
  stlt_prof_add_bytes(4.0 * (double)n);
  int64_t blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, flat_grad, n / 4, n, scratch);
  if (int e = stlt_check_launch("sumsq_kernel")) return e;
  hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(256), 0, s, scratch, (int)blocks, max_norm, out);
  return stlt_check_launch("norm_finish_kernel");
}

int stlt_adamw_step(const stlt_opt_chunk* chunks_dev, int64_t n_chunks, const float* flat_grad, float* exp_avg, float* exp_avg_sq,
                    const float* norm_and_clip, float lr, float beta1, float beta2, float eps, int64_t step, stlt_stream_t stream) {
  if (!chunks_dev || !flat_grad || !exp_avg || !exp_avg_sq) return stlt_set_error(STLT_EINVAL, "stlt_adamw_step: null pointer");
  if (step < 1 || n_chunks < 0 || n_chunks > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "stlt_adamw_step: bad step / chunk count");
  StltProfScope ps(STLT_K_OPTIM, (hipStream_t)stream);
  stlt_prof_note("adamw chunks=%lld (of up to 16384 elements: gradient, parameter and both moments read, parameter and moments written)", (long long)n_chunks);
  stlt_prof_add_bytes(28.0 * 16384.0 * (double)n_chunks);  // upper bound: the last chunk of a parameter is partly filled
  if (n_chunks == 0) return 0;
  // scalars as torch computes them: python floats (double), rounded to fp32 where they meet the tensors
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float neg_step = (float)(-(double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const float w_lerp = (float)(1.0 - (double)beta1), w_sq = (float)(1.0 - (double)beta2);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, chunks_dev, flat_grad, exp_avg, exp_avg_sq,
                     norm_and_clip, lr, w_lerp, beta2, w_sq, eps, neg_step, bc2_sqrt);
  return stlt_check_launch("adamw_kernel");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Criterion (reference utils/train_inference_utils.py:64-76) with its gradient in one pass: CrossEntropyLoss
// ("something", int64 class labels) or BCEWithLogitsLoss ("action_genome", float multi-hot labels), both with the
// default mean reduction.  One block per clip writes the clip's loss term and its dlogits row; a one-block finish sums
// the terms in clip order (deterministic).
namespace {

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = is_max ? fmaxf(red[threadIdx.x], red[threadIdx.x + o]) : red[threadIdx.x] + red[threadIdx.x + o];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void loss_rows_kernel(const float* __restrict__ logits, const void* __restrict__ labels, int kind,
                                                        int K, float grad_scale, float* __restrict__ row_loss,
                                                        float* __restrict__ dlogits) {
  __shared__ float red[256];
  const int64_t b = blockIdx.x;
  const float* x = logits + b * K;
  float* dx = dlogits + b * K;
  if (kind == STLT_LOSS_CROSS_ENTROPY) {
    int64_t y = static_cast<const int64_t*>(labels)[b];
    if (y == -100) {  // nn.CrossEntropyLoss's default ignore_index: the clip contributes nothing (the finish kernel takes the mean over the others)
      for (int k = threadIdx.x; k < K; k += 256) dx[k] = 0.f;
      if (threadIdx.x == 0) row_loss[b] = 0.f;
      return;
    }
    if (y < 0 || y >= K) {  // torch raises on an out-of-range class index; here the row's loss and gradient become NaN, which
      const float nan = __builtin_nanf("");  // the mean loss, the gradient norm and every later step then show
      for (int k = threadIdx.x; k < K; k += 256) dx[k] = nan;
      if (threadIdx.x == 0) row_loss[b] = nan;
      return;
    }
    float m = -INFINITY;
    for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, x[k]);
    m = block_reduce(m, red, true);
    float sum = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) sum += expf(x[k] - m);
    sum = block_reduce(sum, red, false);
    const float lse = m + logf(sum);
    for (int k = threadIdx.x; k < K; k += 256) dx[k] = (expf(x[k] - lse) - (k == y ? 1.0f : 0.0f)) * grad_scale;
    if (threadIdx.x == 0) row_loss[b] = lse - x[y];
  } else {
    const float* y = static_cast<const float*>(labels) + b * K;
    float sum = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
      const float v = x[k], t = y[k];
      sum += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));  // numerically stable BCE-with-logits term
      dx[k] = (1.0f / (1.0f + expf(-v)) - t) * grad_scale;
    }
    sum = block_reduce(sum, red, false);
    if (threadIdx.x == 0) row_loss[b] = sum;
  }
}

// scale = weight / (number of terms the mean runs over, had no clip been ignored).  Cross entropy with ignored clips
// (label -100): the mean runs over the other clips, so the loss and — the rare case, one block walks the rows — the
// gradient rows already written with 1/B are rescaled by B / n_valid; no valid clip at all gives NaN, as torch's 0/0 does.
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ row_loss, int64_t B, float scale, float* __restrict__ out,
                                                          const int64_t* __restrict__ ce_labels, int K, float* __restrict__ dlogits) {
  __shared__ double red[256];
  __shared__ int ign[256];
  double acc = 0.0;
  int n_ign = 0;
  for (int64_t i = threadIdx.x; i < B; i += 256) {
    acc += (double)row_loss[i];
    if (ce_labels && ce_labels[i] == -100) ++n_ign;
  }
  red[threadIdx.x] = acc;
  ign[threadIdx.x] = n_ign;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; ign[threadIdx.x] += ign[threadIdx.x + o]; }
    __syncthreads();
  }
  const int ignored = ign[0];
  const double fix = ignored == 0 ? 1.0 : (double)B / (double)(B - ignored);  // inf when every clip is ignored: 0 * inf = NaN below
  if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)scale * fix);
  if (ignored > 0 && ignored < B) {
    const float f = (float)fix;
    for (int64_t i = threadIdx.x; i < B * K; i += 256) dlogits[i] *= f;
  }
}

}  // namespace

extern "C" int stlt_loss_fwd_bwd(const float* logits, const void* labels, int kind, int64_t B, int64_t K, float weight,
                                 float* scratch, float* loss_out, float* dlogits, stlt_stream_t stream) {
  if (!logits || !labels || !scratch || !loss_out || !dlogits) return stlt_set_error(STLT_EINVAL, "stlt_loss_fwd_bwd: null pointer");
  if (kind != STLT_LOSS_CROSS_ENTROPY && kind != STLT_LOSS_BCE_WITH_LOGITS) return stlt_set_error(STLT_EINVAL, "stlt_loss_fwd_bwd: unknown loss %d", kind);
  if (B <= 0 || K <= 0 || K > 0x7fffffffLL) return stlt_set_error(STLT_EINVAL, "stlt_loss_fwd_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  StltProfScope ps(STLT_K_OPTIM, s);
  // mean reduction: over the clips (cross entropy) or over all B*K elements (BCE)
  const float mean = kind == STLT_LOSS_CROSS_ENTROPY ? 1.0f / (float)B : 1.0f / ((float)B * (float)K);
  hipLaunchKernelGGL(loss_rows_kernel, dim3((unsigned)B), dim3(256), 0, s, logits, labels, kind, (int)K, weight * mean, scratch, dlogits);
  if (int e = stlt_check_launch("loss_rows_kernel")) return e;
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, scratch, B, weight * mean, loss_out,
                     kind == STLT_LOSS_CROSS_ENTROPY ? static_cast<const int64_t*>(labels) : nullptr, (int)K, dlogits);
  return stlt_check_launch("loss_finish_kernel");
}
