// Wave-wide sums on the DPP / permlane data paths of gfx950 (included by rowwise.hip, backward.hip, mhsa.hip, the attention kernels and tools/wave_sum_check.hip).
#pragma once
#include <hip/hip_runtime.h>

// The butterfly sum of common.h's wave_sum (v += v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1) without the LDS crossbar: gfx950's
// v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and even rows of two registers, row_ror:8 is lane ^ 8 within a row
// of 16, row_ror:4 reads lane ^ 4's VALUE once lanes i and i ^ 8 agree, the quad permutes are lane ^ 2 and lane ^ 1.  Every step adds the
// same two numbers as the butterfly, so the result is the butterfly's bit for bit (tools/wave_sum_check.hip runs both on the GPU); it costs
// 8 VALU instructions instead of 6 x (ds_bpermute + 5).  The swaps are inline assembly: the compiler folds the builtin called with one
// value in both operands into v + v (ROCm 7.2).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// own value and the value of lane ^ 16 / lane ^ 32, in some order (enough for max and for a two-term sum, which are symmetric): after the
// swap of two copies, one register holds the own value and the other the partner's in every lane
__device__ __forceinline__ void wave_pair16(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void wave_pair32(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// max / sum over the four 16-lane groups of a wave (lanes ^ 16, ^ 32): what the attention kernels' softmax needs per query column; the same
// values, added / compared in the same pairs, as `v = op(v, __shfl_xor(v, 16)); v = op(v, __shfl_xor(v, 32))`, without the two LDS round trips
#ifndef STLT_GROUPS_SWAP
#define STLT_GROUPS_SWAP 1  // 0: through ds_bpermute shuffles (A/B builds)
#endif
__device__ __forceinline__ float groups_max(float v) {
#if !STLT_GROUPS_SWAP
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
#endif
  float a, b;
  wave_pair16(v, a, b);
  v = fmaxf(a, b);
  wave_pair32(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float groups_sum(float v) {
#if !STLT_GROUPS_SWAP
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
#endif
  float a, b;
  wave_pair16(v, a, b);
  v = a + b;
  wave_pair32(v, a, b);
  return a + b;
}
#ifndef STLT_LN_DPP
#define STLT_LN_DPP 1  // 0: the LayerNorm reductions through common.h's wave_sum (A/B builds)
#endif
__device__ __forceinline__ float wave_sum_dpp(float v) {
#if !STLT_LN_DPP
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  v = a + b;
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  v = a + b;
  v += dpp_mov<0x128>(v);  // row_ror:8
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x4E>(v);   // quad_perm:[2,3,0,1]
  v += dpp_mov<0xB1>(v);   // quad_perm:[1,0,3,2]
  return v;
#endif
}
