// A stand-in HIP runtime for the `--offload-host-only` sanitizer build (tests/test_host_sanitizers.py): the 20 runtime functions the library
// calls, with no device behind them.  Launches "succeed" after their configuration has been checked the way the real runtime checks it
// (non-empty grid, at most 1024 threads per workgroup, grid dimensions below 2^31, at most 160 KB of dynamic LDS), so every entry point runs
// its WHOLE host-side sequence — all the plan arithmetic of a forward or a reverse sweep, not just up to the first launch — and a plan that
// computes an impossible launch is reported as an error return, not hidden.  Nothing here touches the (fake) device pointers it is handed;
// device-to-host copies fill the host destination with zeros.  Defined in the executable, these take precedence over libamdhip64's.
#include <hip/hip_runtime_api.h>

#include <sanitizer/common_interface_defs.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
thread_local hipError_t t_last = hipSuccess;
struct Cfg { dim3 g, b; size_t sh; hipStream_t s; };
thread_local Cfg t_cfg[8];
thread_local int t_depth = 0;
std::atomic<long> g_launches{0}, g_refused{0};
hipError_t fail(hipError_t e) { t_last = e; return e; }
int g_dummy_stream, g_dummy_event;
}  // namespace

extern "C" {
long stlt_fake_hip_launches() { return g_launches.load(); }
long stlt_fake_hip_refused() { return g_refused.load(); }

void** __hipRegisterFatBinary(const void*) { static void* handle = nullptr; return &handle; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned int, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, unsigned long, int, int) {}
void __hipRegisterManagedVar(void*, void**, void*, const char*, unsigned long, unsigned) {}

hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t s) {
  if (t_depth < 8) t_cfg[t_depth] = Cfg{g, b, sh, s};
  ++t_depth;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* s) {
  if (t_depth > 0) --t_depth;
  const Cfg& c = t_cfg[t_depth < 8 ? t_depth : 7];
  *g = c.g; *b = c.b; *sh = c.sh; *s = c.s;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t sh, hipStream_t) {
  const unsigned long long threads = (unsigned long long)b.x * b.y * b.z;
  if (!f || !args || g.x == 0 || g.y == 0 || g.z == 0 || threads == 0 || threads > 1024 || g.x > 0x7fffffffu || g.y > 65535u || g.z > 65535u || sh > 160 * 1024) {
    // STLT_FAKE_HIP_TRACE=n: say where the first n impossible launches came from (the library should have refused the shape itself)
    static const long trace = getenv("STLT_FAKE_HIP_TRACE") ? atol(getenv("STLT_FAKE_HIP_TRACE")) : 0;
    if (g_refused++ < trace) {
      fprintf(stderr, "[fake hip] refused launch: grid (%u,%u,%u) block (%u,%u,%u) lds %zu\n", g.x, g.y, g.z, b.x, b.y, b.z, sh);
      __sanitizer_print_stack_trace();
    }
    return fail(hipErrorInvalidConfiguration);
  }
  ++g_launches;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { const hipError_t e = t_last; t_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorInvalidConfiguration ? "invalid configuration argument" : "fake runtime error"); }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof(*p));
  p->multiProcessorCount = 256;
  p->warpSize = 64;
  p->sharedMemPerBlock = 64 * 1024;
  p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  p->maxThreadsPerBlock = 1024;
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int v) { return v < 0 || v > 160 * 1024 ? fail(hipErrorInvalidValue) : hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int block, size_t lds) {
  *n = lds > 80 * 1024 ? 1 : (block > 512 ? 2 : 4);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* p, int, size_t, hipStream_t) { return p ? hipSuccess : fail(hipErrorInvalidValue); }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t) {
  if (!dst || !src) return fail(hipErrorInvalidValue);
  if (kind == hipMemcpyDeviceToHost) std::memset(dst, 0, n);  // the only host memory a copy may write
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : fail(hipErrorInvalidHandle); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)&g_dummy_stream; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)&g_dummy_event; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)&g_dummy_event; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { return e ? hipSuccess : fail(hipErrorInvalidHandle); }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
}
