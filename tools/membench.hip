// Diagnostic (not part of the product): what HBM rate do the attention kernel's access patterns reach on gfx950?
//   A: every wave streams a contiguous 24 KB block (16 B/lane)           -- "head-major" layout
//   B: every wave reads 3 x (32 rows x 256 B) at a 9216-B row stride      -- packed (tokens, 3d) layout, one head
//   each followed (w=1) or not (w=0) by writing 8 KB (32 rows x 256 B at 3072-B stride / contiguous)
// hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o gpurun_out/membench && gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int WRITE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, long n_items, int H) {
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= n_items) return;
  const int head = item % H; const long tile = item / H;
  f32x4 v[24];
  if (PATTERN == 0) {
    const float* src = in + item * 6144;  // 24 KB contiguous
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (i * 64 + lane) * 4);
  } else {
    const long ld = 3L * H * 64;
    const float* src = in + tile * 32 * ld + head * 64;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = lane + 64 * i, row = f >> 4, c4 = (f & 15) * 4;
        v[t * 8 + i] = *reinterpret_cast<const f32x4*>(src + t * H * 64 + row * ld + c4);
      }
  }
  f32x4 s = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 24; ++i) s += v[i];
  if (WRITE) {
    if (PATTERN == 0) {
      float* dst = out + item * 2048;
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(dst + (i * 64 + lane) * 4) = s + v[i];
    } else {
      const long ldo = (long)H * 64;
      float* dst = out + tile * 32 * ldo + head * 64;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = lane + 64 * i, row = f >> 4, c4 = (f & 15) * 4;
        *reinterpret_cast<f32x4*>(dst + row * ldo + c4) = s + v[i];
      }
    }
  } else if (s.x == 12345.678f) out[item] = s.y;
}

template <int P, int W>
void run(const char* name, const float* in, float* out, long n_items, int H) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  dim3 grid((unsigned)((n_items + 3) / 4));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<P, W>), grid, dim3(256), 0, 0, in, out, n_items, H);
  hipEventRecord(a);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<P, W>), grid, dim3(256), 0, 0, in, out, n_items, H);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
  double bytes = n_items * (24576.0 + (W ? 8192.0 : 0.0));
  printf("%-28s items=%ld  %8.1f us  %7.1f GB/s\n", name, n_items, ms * 1e3, bytes / ms / 1e6);
}

int main(int argc, char** argv) {
  const int H = 12; const long S = argc > 1 ? atol(argv[1]) : 1024; const long n_items = S * H;
  float *in, *out;
  hipMalloc(&in, n_items * 24576); hipMalloc(&out, n_items * 8192);
  hipMemset(in, 0, n_items * 24576); hipMemset(out, 0, n_items * 8192);
  run<0, 0>("A contiguous  read", in, out, n_items, H);
  run<0, 1>("A contiguous  read+write", in, out, n_items, H);
  run<1, 0>("B strided     read", in, out, n_items, H);
  run<1, 1>("B strided     read+write", in, out, n_items, H);
  return 0;
}
