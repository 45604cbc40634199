// Evaluators on the device: the counting / ranking arithmetic of the reference's
// src/utils/evaluation.py as HIP kernels, fed straight from the logits the forward leaves in HBM.
//
//   eval_topk_kernel      EvaluatorSomething.process (evaluation.py:21-34): per clip, is the label the arg-max, and is
//                         it among the five largest logits.  One wave per clip: the label's rank is the number of
//                         classes that beat it (larger logit, or equal logit at a lower index — the order a stable
//                         descending sort gives; for k = 1 that is torch.argmax's first-maximum rule), counted with
//                         one ballot per 64 classes.  Hits go to two int64 device counters (integer atomics:
//                         order-free, so reproducible).
//   eval_empty_rows_kernel charades_map's "clips without any ground-truth action sort last" (evaluation.py:127-131).
//   eval_ap_kernel        map() (evaluation.py:100-124): one workgroup per class sorts the clips of its column by
//                         descending score (bitonic sort of 64-bit keys in LDS, clip index as tie-break = stable),
//                         then precision at every positive from a blocked prefix count, summed in clip order per
//                         thread and in thread order across the block (float64, fixed order: reproducible).
#include "common.h"

namespace {

constexpr int EV_THREADS = 256;
constexpr int EV_MAX_CLIPS = 16384;  // 8-byte keys in LDS: 128 KB

__global__ __launch_bounds__(256) void eval_topk_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                        int64_t B, int K, unsigned long long* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const int64_t y = labels[row];
  if (y < 0 || y >= K) return;  // an out-of-range label can never be hit (torch would compare it against indices < K too)
  const float* x = logits + row * ld;
  const float xy = x[y];
  int beaten_by = 0;
  for (int j0 = 0; j0 < K; j0 += 64) {
    const int j = j0 + lane;
    bool beats = false;
    if (j < K) {
      const float v = x[j];
      beats = v > xy || (v == xy && j < (int)y);
    }
    beaten_by += __popcll(__ballot(beats));
  }
  if (lane == 0) {
    if (beaten_by < 1) atomicAdd(&counts[0], 1ull);
    if (beaten_by < 5) atomicAdd(&counts[1], 1ull);
  }
}

__global__ __launch_bounds__(256) void eval_empty_rows_kernel(const float* __restrict__ truths, int64_t n, int C, uint8_t* __restrict__ empty) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  bool any = false;
  for (int j = lane; j < C; j += 64) any |= truths[row * C + j] != 0.f;
  const unsigned long long seen = __ballot(any);  // all 64 lanes vote before lane 0 writes
  if (lane == 0) empty[row] = seen == 0ull ? 1 : 0;
}

// monotone map float -> uint32 (larger float = larger key), then inverted so that an ASCENDING key sort is a
// DESCENDING score sort; -inf becomes the largest key
__device__ __forceinline__ uint32_t desc_key(float v) {
  uint32_t u = __float_as_uint(v);
  u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
  return ~u;
}

__global__ __launch_bounds__(EV_THREADS) void eval_ap_kernel(const float* __restrict__ scores, const float* __restrict__ truths,
                                                             const uint8_t* __restrict__ empty, int n, int n_pow2, int C,
                                                             double* __restrict__ ap_out, double* __restrict__ pos_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // n_pow2 keys, then EV_THREADS counts + sums
  const int c = blockIdx.x, tid = threadIdx.x;
  // key = (inverted score bits) << 32 | clip << 1 | truth; padding keys are all ones and sort behind everything
  for (int i = tid; i < n_pow2; i += EV_THREADS) {
    unsigned long long k = ~0ull;
    if (i < n) {
      const float s = empty[i] ? -__builtin_inff() : scores[(int64_t)i * C + c];
      const unsigned t = truths[(int64_t)i * C + c] == 1.f ? 1u : 0u;
      k = ((unsigned long long)desc_key(s) << 32) | ((unsigned long long)(unsigned)i << 1) | t;
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int size = 2; size <= n_pow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (n_pow2 >> 1); i += EV_THREADS) {
        const int lo = 2 * i - (i & (stride - 1));  // index of the lower element of the i-th pair at this stride
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  // blocked prefix count of the positives: thread t owns ranks [t*chunk, (t+1)*chunk)
  unsigned* cnt = reinterpret_cast<unsigned*>(keys + n_pow2);
  double* part = reinterpret_cast<double*>(cnt + EV_THREADS);
  const int chunk = (n + EV_THREADS - 1) / EV_THREADS;
  const int r0 = tid * chunk, r1 = r0 + chunk < n ? r0 + chunk : n;
  unsigned mine = 0;
  for (int r = r0; r < r1; ++r) mine += (unsigned)(keys[r] & 1ull);
  cnt[tid] = mine;
  __syncthreads();
  unsigned before = 0;
  for (int t = 0; t < tid; ++t) before += cnt[t];
  double sum = 0.0;
  unsigned tpc = before;
  for (int r = r0; r < r1; ++r) {
    if (keys[r] & 1ull) {
      ++tpc;
      sum += (double)tpc / (double)(r + 1);  // t_pcs / (f_pcs + t_pcs): the denominator is the 1-based rank
    }
  }
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    unsigned n_pos = 0;
    for (int t = 0; t < EV_THREADS; ++t) { tot += part[t]; n_pos += cnt[t]; }
    ap_out[c] = n_pos > 0 ? tot / (double)n_pos : __builtin_nan("");
    pos_out[c] = (double)n_pos;
  }
}

}  // namespace

namespace {
// one thread per (clip, class): the fp32 sigmoid of the logit (what `x.float().sigmoid()` computes) and the label, widened to
// the float64 tables the mAP is computed from
__global__ __launch_bounds__(256) void eval_store_sigmoid_kernel(const float* __restrict__ logits, int64_t ld, const float* __restrict__ labels,
                                                                 int64_t B, int C, double* __restrict__ pred, double* __restrict__ truth) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int64_t r = i / C;
  const int c = (int)(i - r * C);
  const float x = logits[r * ld + c];
  pred[i] = (double)(1.0f / (1.0f + expf(-x)));
  truth[i] = (double)labels[i];
}
}  // namespace

extern "C" {

int stlt_eval_store_sigmoid(const float* logits, int64_t ld, const float* labels, int64_t B, int64_t C, double* pred, double* truth,
                            int64_t row0, stlt_stream_t stream) {
  if (!logits || !labels || !pred || !truth) return stlt_set_error(STLT_EINVAL, "stlt_eval_store_sigmoid: null pointer");
  if (B < 0 || C <= 0 || C > 0x7fffffff || ld < C || row0 < 0 || B > (int64_t)0x7fffffff * 256 / C || row0 > ((int64_t)1 << 40) / C)
    return stlt_set_error(STLT_EINVAL, "stlt_eval_store_sigmoid: bad shape");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  StltProfScope ps(STLT_K_MISC, s);
  hipLaunchKernelGGL(eval_store_sigmoid_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, s, logits, ld, labels, B, (int)C,
                     pred + row0 * C, truth + row0 * C);
  return stlt_check_launch("eval_store_sigmoid_kernel");
}

int stlt_eval_topk(const float* logits, int64_t ld, const int64_t* labels, int64_t B, int64_t K, int64_t* counts,
                   stlt_stream_t stream) {
  if (B == 0) return 0;  // an empty batch has no storage to point at
  if (!logits || !labels || !counts) return stlt_set_error(STLT_EINVAL, "stlt_eval_topk: null pointer");
  if (B < 0 || K <= 0 || K > 0x7fffffff || ld < K) return stlt_set_error(STLT_EINVAL, "stlt_eval_topk: bad shape (B=%lld, K=%lld, ld=%lld)", (long long)B, (long long)K, (long long)ld);
  hipLaunchKernelGGL(eval_topk_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, B, (int)K,
                     reinterpret_cast<unsigned long long*>(counts));
  return stlt_check_launch("eval_topk_kernel");
}

int64_t stlt_eval_max_clips(void) { return EV_MAX_CLIPS; }

int stlt_eval_average_precision(const float* scores, const float* truths, int64_t n, int64_t C, double* ap, double* positives,
                                uint8_t* scratch, stlt_stream_t stream) {
  if (!scores || !truths || !ap || !positives || !scratch) return stlt_set_error(STLT_EINVAL, "stlt_eval_average_precision: null pointer");
  if (n <= 0 || C <= 0 || C > 0x7fffffff) return stlt_set_error(STLT_EINVAL, "stlt_eval_average_precision: bad shape (n=%lld, C=%lld)", (long long)n, (long long)C);
  if (n > EV_MAX_CLIPS)
    return stlt_set_error(STLT_EINVAL, "stlt_eval_average_precision: %lld clips exceed the %d a class column is sorted with in LDS", (long long)n, EV_MAX_CLIPS);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(eval_empty_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, truths, n, (int)C, scratch);
  if (int e = stlt_check_launch("eval_empty_rows_kernel")) return e;
  int n_pow2 = 2;
  while (n_pow2 < n) n_pow2 <<= 1;
  const size_t lds = (size_t)n_pow2 * 8 + EV_THREADS * (sizeof(unsigned) + sizeof(double));
  static StltPerDeviceOnce once;  // > 64 KB of dynamic LDS needs the attribute, per device
  bool& opted = once.flag();
  if (!opted) {
    if (hipError_t e = hipFuncSetAttribute((const void*)eval_ap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); e != hipSuccess)
      return stlt_set_error((int)e, "stlt_eval_average_precision: hipFuncSetAttribute: %s", hipGetErrorString(e));
    opted = true;
  }
  hipLaunchKernelGGL(eval_ap_kernel, dim3((unsigned)C), dim3(EV_THREADS), lds, s, scores, truths, scratch, (int)n, n_pow2, (int)C, ap, positives);
  return stlt_check_launch("eval_ap_kernel");
}

}  // extern "C"
