// Device-side collater: the padding half of the reference's StltCollater.__call__
// (src/modelling/datasets.py:243-288 with pad_sequence, src/utils/data_utils.py:93-102) as one kernel.
// Input: the per-video layouts of StltDataset.__getitem__ (datasets.py:52-125) concatenated along the frame axis
// (ragged: video b owns frames [offsets[b], offsets[b+1])).  Output: the padded (B,T,N,.) batch, both key-padding
// masks and nothing else — padded frames carry the CLS object in slot 0 (category = cls id, box [0,0,1,1], score 1)
// and frame type "pad" = 0, exactly like the reference's pad tensors.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void collate_kernel(const int64_t* __restrict__ cat_r, const float* __restrict__ box_r,
                                                      const float* __restrict__ score_r, const int64_t* __restrict__ ft_r,
                                                      const int64_t* __restrict__ offsets, int64_t B, int T, int N,
                                                      int64_t cls_id, int64_t* __restrict__ cat, float* __restrict__ box,
                                                      float* __restrict__ score, int64_t* __restrict__ ft,
                                                      uint8_t* __restrict__ kpm_boxes, uint8_t* __restrict__ kpm_frames) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (b, t, n)
  if (idx >= B * T * N) return;
  const int n = (int)(idx % N);
  const int64_t bt = idx / N;
  const int t = (int)(bt % T);
  const int64_t b = bt / T;
  const int64_t f0 = offsets[b], len = offsets[b + 1] - f0;
  int64_t c;
  f32x4 bx;
  float sc;
  if (t < len) {
    const int64_t src = (f0 + t) * N + n;
    c = cat_r[src];
    bx = *reinterpret_cast<const f32x4*>(box_r + src * 4);
    sc = score_r ? score_r[src] : 0.f;
  } else {  // pad_categories_tensor / pad_boxes_tensor / pad_scores_tensor (datasets.py:247-264)
    c = n == 0 ? cls_id : 0;
    bx = n == 0 ? f32x4{0.f, 0.f, 1.f, 1.f} : f32x4{0.f, 0.f, 0.f, 0.f};
    sc = n == 0 ? 1.f : 0.f;
  }
  cat[idx] = c;
  *reinterpret_cast<f32x4*>(box + idx * 4) = bx;
  if (score) score[idx] = sc;
  kpm_boxes[idx] = c == 0;  // src_key_padding_mask_boxes = categories == 0 (datasets.py:274-278)
  if (n == 0) {
    const int64_t v = t < len ? ft_r[f0 + t] : 0;  // frame2type["pad"] = 0 (configs.py:79-89)
    ft[bt] = v;
    kpm_frames[bt] = v == 0;  // datasets.py:280-286
  }
}

}  // namespace

extern "C" int stlt_collate_fwd(const int64_t* categories_ragged, const float* boxes_ragged, const float* scores_ragged,
                                const int64_t* frame_types_ragged, const int64_t* frame_offsets, int64_t B, int64_t T,
                                int64_t N, int64_t cls_id, int64_t* categories, float* boxes, float* scores,
                                int64_t* frame_types, uint8_t* kpm_boxes, uint8_t* kpm_frames, stlt_stream_t stream) {
  if (!categories_ragged || !boxes_ragged || !frame_types_ragged || !frame_offsets || !categories || !boxes || !frame_types ||
      !kpm_boxes || !kpm_frames)
    return stlt_set_error(STLT_EINVAL, "stlt_collate_fwd: null pointer");
  if ((scores_ragged == nullptr) != (scores == nullptr))
    return stlt_set_error(STLT_EINVAL, "stlt_collate_fwd: scores input and output must be given together");
  if (B < 0 || T <= 0 || N <= 0) return stlt_set_error(STLT_EINVAL, "stlt_collate_fwd: bad shape");
  if (B == 0) return 0;
  const int64_t n = B * T * N;
  hipLaunchKernelGGL(collate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, categories_ragged,
                     boxes_ragged, scores_ragged, frame_types_ragged, frame_offsets, B, (int)T, (int)N, cls_id, categories, boxes,
                     scores, frame_types, kpm_boxes, kpm_frames);
  return stlt_check_launch("collate_kernel");
}
