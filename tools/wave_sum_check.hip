// The DPP / permlane wave sum of csrc/wave_dpp.h against the ds_bpermute butterfly it replaces, bit for bit, on the GPU:
//   hipcc -O3 --offload-arch=gfx950 -I revisiting-spatial-temporal-layouts_amd/csrc tools/wave_sum_check.hip -o /tmp/wave_sum_check && /tmp/wave_sum_check
// (tests/test_kernels_gpu.py builds and runs it).  Exit code 0: every lane of every wave agrees.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "wave_dpp.h"

__device__ __forceinline__ float butterfly_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// mode 0: the 64-lane sum; mode 1: max over lanes ^ 16, ^ 32 (mhsa.hip's softmax reductions on wave_pair16 / wave_pair32); mode 2: that sum
__global__ __launch_bounds__(256) void both_kernel(const float* __restrict__ in, float* __restrict__ dpp, float* __restrict__ ref, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float v = in[i];
  if (mode == 0) {
    dpp[i] = wave_sum_dpp(v);
    ref[i] = butterfly_sum(v);
    return;
  }
  float a, b, x = v, y = v;
  wave_pair16(x, a, b);
  x = mode == 1 ? fmaxf(a, b) : a + b;
  wave_pair32(x, a, b);
  x = mode == 1 ? fmaxf(a, b) : a + b;
  dpp[i] = x;
  y = mode == 1 ? fmaxf(y, __shfl_xor(y, 16, 64)) : y + __shfl_xor(y, 16, 64);
  y = mode == 1 ? fmaxf(y, __shfl_xor(y, 32, 64)) : y + __shfl_xor(y, 32, 64);
  ref[i] = y;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main() {
  const int n = 256 * 4096;
  std::vector<float> h(n), a(n), b(n);
  unsigned s = 12345u;
  for (int i = 0; i < n; ++i) {  // mixed magnitudes and signs: rounding differs between summation orders
    s = s * 1664525u + 1013904223u;
    const float u = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 20));
    h[i] = u * (float)(1 << (i % 11)) * ((i / 64) % 3 == 0 ? 1e-3f : 1.0f);
  }
  float *d_in, *d_a, *d_b;
  CHECK(hipMalloc(&d_in, n * 4)); CHECK(hipMalloc(&d_a, n * 4)); CHECK(hipMalloc(&d_b, n * 4));
  CHECK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
  long bad = 0, uniform_bad = 0;
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(both_kernel, dim3(n / 256), dim3(256), 0, 0, d_in, d_a, d_b, mode);
    CHECK(hipGetLastError());
    CHECK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
      if (memcmp(&a[i], &b[i], 4)) { if (bad < 5) printf("mode %d lane %d: swap/dpp %.9g shuffle %.9g\n", mode, i, a[i], b[i]); ++bad; }
      if (mode == 0 && memcmp(&a[i], &a[i & ~63], 4)) ++uniform_bad;  // every lane of a wave holds the same sum
    }
  }
  printf("{\"values\": %d, \"mismatches\": %ld, \"lanes_disagreeing_within_a_wave\": %ld}\n", n, bad, uniform_bad);
  return (bad || uniform_bad) ? 1 : 0;
}
