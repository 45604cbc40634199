// Shared device/host helpers for libstlt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stlt_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define STLT_WAVE 64

// error plumbing (api.hip)
int stlt_set_error(int code, const char* fmt, ...);
int stlt_check_launch(const char* what);

// Per-device state (api.hip).  Nothing in the library caches a property of "the first device it saw": CU counts,
// occupancy answers and one-time function attributes are kept per HIP device, looked up by the device that is current
// when a launcher runs (one process may drive several GPUs).
constexpr int STLT_MAX_DEVICES = 64;
int stlt_current_device();           // hipGetDevice; 0 if the query fails
int stlt_device_cus();               // compute units of the current device (256 if the query fails)
struct StltPerDeviceOnce {           // "has this one-time step run on the current device yet?"
  bool done[STLT_MAX_DEVICES] = {};
  bool& flag() { return done[stlt_current_device() & (STLT_MAX_DEVICES - 1)]; }
};
struct StltPerDeviceInt {            // small per-device cache (0 = not yet known)
  int v[STLT_MAX_DEVICES] = {};
  int& ref() { return v[stlt_current_device() & (STLT_MAX_DEVICES - 1)]; }
};

// diagnostics buffer (api.hip: stlt_debug_set_buffer); NULL in normal operation
extern unsigned long long* g_stlt_debug_buf;

// per-kernel timing hooks (api.hip)
void stlt_prof_begin(int kid, hipStream_t s);
void stlt_prof_end(int kid, hipStream_t s);
void stlt_prof_add_flops(double flops);
void stlt_prof_note(const char* fmt, ...) __attribute__((format(printf, 1, 2)));  // the open scope's launch in words (shape, tile, workgroups, rounds, k-steps)
void stlt_prof_add_bytes(double bytes);
void stlt_prof_note_flops(double flops);                                         // ... and FLOPs that do not belong to the GEMM roofline's sum (stlt_prof_take_gemm_flops)                                          // ... and its algorithmic bytes (HBM-bound kernels)
struct StltProfScope {  // scopes nest: only the outermost one of a thread records (a launcher that calls another launcher is one entry)
  int kid; hipStream_t s;
  StltProfScope(int k, hipStream_t st) : kid(k), s(st) { stlt_prof_begin(kid, s); }
  ~StltProfScope() { stlt_prof_end(kid, s); }
};

// One LDS-DMA instruction in its scalar-base form: 16 (4) bytes per lane from (wave-uniform 64-bit base in SGPRs) + (per-lane unsigned 32-bit
// byte offset) to LDS byte address lds_addr + 16 (4) * lane.  hipcc's builtin takes a 64-bit per-lane pointer and only sometimes folds
// base + zext(offset) back into this form (it keeps zero-extended offsets in register pairs and adds the base with v_lshl_add_u64 per
// instruction: a vector-ALU instruction beside the MFMA waves for every DMA).  The instruction is invisible to the compiler's vmcnt
// bookkeeping: callers wait with explicit s_waitcnt vmcnt(N), as the loader waves do anyway; M0 is declared clobbered (hipcc warns that it is a
// reserved register: a wave that uses these helpers issues none of its LDS-DMA through the builtin, so nothing else of it lives in M0).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void stlt_dma16(const void* uniform_base, uint32_t lane_byte_offset, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_byte_offset), "s"(uniform_base), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void stlt_dma4(const void* uniform_base, uint32_t lane_byte_offset, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(lane_byte_offset), "s"(uniform_base), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ uint32_t stlt_lds_addr(const void* lds_ptr) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)lds_ptr;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

#ifndef STLT_GELU_BRANCH_FREE
#define STLT_GELU_BRANCH_FREE 1
#endif
// GELU of the FFN1 epilogue.  The library erff takes one of two branches per lane (|z| < 1: 8 instructions; else a
// degree-7 polynomial + exp, ~24), so its cost in a 64-lane wave depends on the data: cheap while every lane of a wave
// is below 1, ~38 instructions once both branches are live.  STLT_GELU_BRANCH_FREE=1 selects a fixed-cost form instead:
// erf(t) = 1 - 2^q(t) for t = min(|z|, 3.95) with q a degree-11 fit of log2(erfc) and one v_exp_f32; max abs error of
// erf 1.1e-7 in fp32 (200k points), the order of erff's own rounding.
__device__ __forceinline__ float gelu_epilogue(float x) {
#if STLT_GELU_BRANCH_FREE
  const float z = x * 0.70710678118654752440f;
  const float t = fminf(fabsf(z), 3.95f);
  float q = 1.1830035617776957e-07f;
  q = fmaf(q, t, -3.0875787615514128e-06f);
  q = fmaf(q, t, 3.5860794014297426e-05f);
  q = fmaf(q, t, -0.00024206875241361558f);
  q = fmaf(q, t, 0.0010191010078415275f);
  q = fmaf(q, t, -0.002435620641335845f);
  q = fmaf(q, t, 0.00011764218652388081f);
  q = fmaf(q, t, 0.027792135253548622f);
  q = fmaf(q, t, -0.14836618304252625f);
  q = fmaf(q, t, -0.9184255599975586f);
  q = fmaf(q, t, -1.6279090642929077f);
  q = fmaf(q, t, 2.831300349726007e-08f);
  const float e = copysignf(1.0f - __builtin_amdgcn_exp2f(q), z);
  return 0.5f * x * (1.0f + e);
#else
  return gelu_erf(x);
#endif
}


// Counter-based dropout: keep element `idx` of site `site` iff mix32(idx; key(seed, site)) >= thr (thr = p * 2^32).
// No mask is stored: the backward recomputes it.  Sites per encoder layer g (0.. spatial, then temporal):
// 8g+0 attention probabilities, 8g+1 after out-proj, 8g+2 FFN hidden, 8g+3 after linear2; 0xE0 / 0xE1 = embedding outputs.
// key = splitmix64 finaliser of seed*G1 + site*G2 (wave-uniform: scalar ALU); the per-element mixer is a keyed
// two-round 32-bit multiply-xorshift (round-3 change: the 64-bit splitmix per element cost ~180 VALU cycles, which is
// exposed once the mask is applied inside a GEMM epilogue; this one is ~55).  oracle/stlt_oracle.py restates it.
struct StltDrop { uint32_t thr; float scale; uint64_t seed; };  // thr == 0: dropout off
__device__ __forceinline__ uint64_t stlt_drop_key(const StltDrop& dr, uint32_t site) {
  uint64_t z = dr.seed * 0x9E3779B97F4A7C15ull + (uint64_t)site * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ bool stlt_keep_k(uint32_t thr, uint64_t key, uint64_t idx) {
  uint32_t x = (uint32_t)idx + (uint32_t)key + (uint32_t)(idx >> 32) * 0x9E3779B9u;
  x = (x ^ (x >> 16)) * 0x7FEB352Du;
  x ^= (uint32_t)(key >> 32);
  x = (x ^ (x >> 15)) * 0x846CA68Bu;
  x ^= x >> 16;
  return x >= thr;
}
__device__ __forceinline__ bool stlt_keep(const StltDrop& dr, uint32_t site, uint64_t idx) {
  return stlt_keep_k(dr.thr, stlt_drop_key(dr, site), idx);
}
__device__ __forceinline__ f32x4 stlt_drop4(const StltDrop& dr, uint32_t site, uint64_t idx0, f32x4 v) {
  const uint64_t key = stlt_drop_key(dr, site);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = stlt_keep_k(dr.thr, key, idx0 + k) ? v[k] * dr.scale : 0.f;
  return o;
}
inline StltDrop stlt_drop_make(float p, uint64_t seed) {
  StltDrop d{0u, 1.0f, seed};
  if (p > 0.f) { double t = (double)p * 4294967296.0; d.thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t; d.scale = 1.0f / (1.0f - p); }
  return d;
}
#define STLT_SITE_EMBED 0xE0u
#define STLT_SITE_FRAMES 0xE1u

// internal launchers shared between the per-kernel C-ABI and the whole-path entry points
int launch_embed(const int64_t* categories, const float* boxes, const float* scores, const float* cat_table,
                 int64_t n_categories, const float* box_w, const float* box_b, const float* score_w,
                 const float* score_b, const float* ln_w, const float* ln_b, float eps, int64_t n_tokens, int64_t d,
                 float* out, hipStream_t s, float* pre_out = nullptr, StltDrop dr = StltDrop{0u, 1.0f, 0ull},
                 const int* src_index = nullptr);
int launch_linear(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int64_t M,
                  int64_t N, int64_t K, int act, hipStream_t s);
int launch_linear_add(const float* x, int64_t ldx, const float* w, const float* bias, const float* r, int64_t ldr, float* y,
                      int64_t ldy, int64_t M, int64_t N, int64_t K, hipStream_t s);
// u (M, N) = x·Wᵀ + b and h (M, N) = drop(gelu(u)) (dropout of site `site`, indices of rows drop_rows[] when given: launch_gelu_fwd's
// arguments): one small-tile launch with the STLT_ACT_GELU_KEEP epilogue when the routing takes the product, else product + launch_gelu_fwd
int launch_linear_gelu_keep(const float* x, int64_t ldx, const float* w, const float* bias, float* u, float* h, int64_t M, int64_t N, int64_t K,
                            StltDrop dr, uint32_t site, const int* drop_rows, hipStream_t s);
// Epilogue of the fused GELU backward (act == STLT_ACT_GELU_BWD, internal): C = drop(A·B) ∘ gelu'(U) with U passed as the
// add-source (r / ldr), the dropout of the FFN hidden (site, drop_rows as launch_gelu_bwd_colsum), and the column sums of C
// (the producing Linear's bias gradient) left as partial rows: cs_part[(tile row * 16 + slot) * N + n], 16 slots per 256-row
// tile row, every slot written — launch_reduce_slabs(cs_part, N, ceil(M/256)*16, g, N, 1) finishes them.
constexpr int STLT_ACT_GELU_BWD = 3;
// Epilogue of the training forward's FFN1 (internal, small-tile kernel only): the pre-activation u = x·W1ᵀ + b1 is stored to the
// add-source pointer (r / ldr, written here, not read) and h = drop(gelu(u)) to the output — the tape keeps both (gelu' needs u, the
// second product and its weight gradient need h); one launch instead of the product + launch_gelu_fwd's read of u.
constexpr int STLT_ACT_GELU_KEEP = 4;
struct StltGemmEpi { StltDrop dr; uint32_t site; const int* drop_rows; float* cs_part; };
// d/dx gelu(x) = Phi(x) + x phi(x), fixed cost like gelu_epilogue (same erf fit, two v_exp_f32)
__device__ __forceinline__ float gelu_grad_epilogue(float x) {
  const float z = x * 0.70710678118654752440f;
  const float t = fminf(fabsf(z), 3.95f);
  float q = 1.1830035617776957e-07f;
  q = fmaf(q, t, -3.0875787615514128e-06f);
  q = fmaf(q, t, 3.5860794014297426e-05f);
  q = fmaf(q, t, -0.00024206875241361558f);
  q = fmaf(q, t, 0.0010191010078415275f);
  q = fmaf(q, t, -0.002435620641335845f);
  q = fmaf(q, t, 0.00011764218652388081f);
  q = fmaf(q, t, 0.027792135253548622f);
  q = fmaf(q, t, -0.14836618304252625f);
  q = fmaf(q, t, -0.9184255599975586f);
  q = fmaf(q, t, -1.6279090642929077f);
  q = fmaf(q, t, 2.831300349726007e-08f);
  const float e = copysignf(1.0f - __builtin_amdgcn_exp2f(q), z);
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  return fmaf(x, pdf, 0.5f * (1.0f + e));
}
// dropout mask of the FFN hidden + gelu'(u) on four consecutive columns of row `drow`
__device__ __forceinline__ f32x4 gelu_bwd4(f32x4 g, f32x4 u, const StltGemmEpi& epi, uint64_t key, uint64_t idx0) {
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = g[j];
    if (epi.dr.thr) v = stlt_keep_k(epi.dr.thr, key, idx0 + j) ? v * epi.dr.scale : 0.f;
    o[j] = v * gelu_grad_epilogue(u[j]);
  }
  return o;
}
// gemm_any.hip: launch_gemm's fallback for contraction lengths that are not multiples of 32 (tiles staged by ordinary loads, same layouts and epilogues)
int launch_gemm_any(int transA, int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, const float* r,
                    int64_t ldr, float* c, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t s);
// gemm_any.hip: products of at most 128 rows as split-k partial tiles + a finishing launch (needs lent stream-K scratch); *taken = launched
int launch_gemm_skinny(int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, const float* r, int64_t ldr, float* c,
                       int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t s, bool* taken);
// gemm_bf16x3.hip: the forward product on the BF16 matrix cores with three-piece operands (opt-in); *taken = launched
float* stlt_gemm_scratch_ptr(size_t* bytes);  // gemm.hip: the calling thread's lent stream-K scratch (nullptr: none)
bool stlt_split_bf16_takes(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw);  // would it (switched on, shape fits)?
int launch_input_grad_bf16x3(const float* dy, int64_t ld_dy, const float* w, int64_t n_out, int64_t k_in, const float* r, int64_t ldr, float* c,
                             int64_t ldc, int64_t rows, float* wt_scratch, hipStream_t s, bool* taken);  // dX = dY·W through a transposed copy of W
int launch_linear_bf16x3(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* r, int64_t ldr, float* y,
                         int64_t ldy, int64_t M, int64_t N, int64_t K, int act, hipStream_t s, bool* taken);
// gemm16.hip: the same product on 128 x (16 NT) whole tiles for under-filled launches (no stream-K, no fix-up); *taken = launched
double stlt_linear_est_us(int64_t M, int64_t N, int64_t K);  // launch-time estimate of launch_linear's duration (us)
int stlt_gemm16_set_mode(int mode);  // -1 by estimate (default), 0 off, 1 always
// ctx.h / wt_cache.hip: the training-loop context of the public call this thread is inside (nullptr: the call named none).  StltCtxScope makes
// `c` current for a call launching on `s` and orders `s` behind a refresh of the context's transposed weight copies that is still in flight.
struct stlt_ctx;
stlt_ctx* stlt_ctx_current();
class StltCtxScope {
 public:
  StltCtxScope(stlt_ctx* c, hipStream_t s);
  ~StltCtxScope();
  StltCtxScope(const StltCtxScope&) = delete;
  StltCtxScope& operator=(const StltCtxScope&) = delete;
  int error() const { return err_; }  // != 0: the handle is not a live context (or the wait could not be enqueued)
 private:
  stlt_ctx* prev_;
  int err_ = 0;
};
bool stlt_wt_lookup(const float* w, int64_t n_out, int64_t k_in, const float** wt, int64_t* ldwt);  // the current context's transposed copy of (a row range of) a weight
void stlt_wt_count_hit();  // a product was launched on a copy
int stlt_gemm16_choice(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, bool wkn = false);  // 0 = gemm.hip keeps the product, else the tile code (row blocks << 5 | column tiles; wkn: the input-gradient build)
int stlt_gemm16_tile_from_public(int tile);  // C-ABI tile parameter (columns | rows << 16, rows 0 = 128) -> tile code, 0 = not a tile of the kernel
int stlt_gemm16_tile_to_public(int code);
int launch_linear_gemm16(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* r, int64_t ldr, float* y,
                         int64_t ldy, int64_t M, int64_t N, int64_t K, int act, hipStream_t s, bool* taken, int force_tile = 0);
int launch_input_grad_gemm16(const float* dy, int64_t ld_dy, const float* w, int64_t n_out, int64_t k_in, const float* r, int64_t ldr, float* c,
                             int64_t ldc, int64_t rows, hipStream_t s, bool* taken, int force_tile = 0, const StltGemmEpi* gelu_bwd = nullptr);  // dX = dY·W on the small tiles, W as it lies
int launch_gemm(int transA, int transB, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                const float* r, int64_t ldr, float* c, int64_t ldc, int64_t slab_stride, int64_t M, int64_t N,
                int64_t K, int n_split, int act, hipStream_t s, const StltGemmEpi* epi = nullptr);
// Scratch for stream-K partial tiles (2 images of 256x128 floats per workgroup).  While a StltGemmScratch is alive
// on the calling thread, launch_gemm may cut under-filled launches into equal k-step ranges; the buffer is only
// touched by kernels enqueued on the launch stream.  Whole-path entry points lend a slice of their workspace.
constexpr int STLT_GEMM_SK_MAX_WG = 256;
constexpr size_t STLT_GEMM_SCRATCH_BYTES = (size_t)2 * STLT_GEMM_SK_MAX_WG * 256 * 128 * sizeof(float);  // 64 MiB
class StltGemmScratch {
 public:
  StltGemmScratch(void* p, size_t bytes);
  ~StltGemmScratch();
  StltGemmScratch(const StltGemmScratch&) = delete;
  StltGemmScratch& operator=(const StltGemmScratch&) = delete;
 private:
  float* prev_;
  size_t prev_bytes_;
};
void stlt_gemm_set_scratch_impl(void* p, size_t bytes);
class StltGemmWgCap {  // while alive on the calling thread: GEMM launches use at most n workgroups (0 = no cap)
 public:
  explicit StltGemmWgCap(int n);
  ~StltGemmWgCap();
  StltGemmWgCap(const StltGemmWgCap&) = delete;
  StltGemmWgCap& operator=(const StltGemmWgCap&) = delete;
 private:
  int prev_;
};
// Grouped launch: several independent products of the same operand layout walked by ONE persistent stream-K launch (one
// fix-up instead of one per product; a workgroup's range may run from one product's tiles into the next one's).  Used for
// the weight gradients of an encoder layer: C_p (M_p, N_p) += A_pᵀ·B_p with A_p (Kc_p, M_p), B_p (Kc_p, N_p) row-major.
constexpr int STLT_GEMM_GROUP_MAX = 32;
struct StltGemmProblem { const float* a; const float* b; const float* r; float* c; int lda, ldb, ldr, ldc; int M, N, nk, tiles_n; };
struct StltGemmGroup {
  int n, pad;
  int tile_base[STLT_GEMM_GROUP_MAX + 1];  // first tile of problem p in the launch's tile order (problem-major, then M, N fastest)
  int step_base[STLT_GEMM_GROUP_MAX + 1];  // first k-step of problem p in the launch's flattened k-step space
  StltGemmProblem p[STLT_GEMM_GROUP_MAX];
};
struct StltWeightGradItem { const float* dy; int64_t n_out; const float* x; int64_t k_in; int64_t rows; float* g_w; };  // g_w (n_out,k_in) += dy[:rows]ᵀ·x[:rows]
int launch_weight_grad_group(const StltWeightGradItem* items, int n_items, hipStream_t s);  // needs lent stream-K scratch (StltGemmScratch); rows % 32 == 0
bool stlt_gemm_has_scratch();
int launch_reduce_slabs(const float* slabs, int64_t stride, int n_slabs, float* dst, int64_t n, int accumulate, hipStream_t s);
// Deferred partial-row reductions: while a StltReduceDefer is set on the calling thread, the accumulating (+=) launch_reduce_slabs /
// launch_reduce_slabs3 calls on its stream are collected and run as ONE batched launch per flush, each destination summed in the same
// fixed order.  The producers' partial rows must then outlive the call that wrote them: stlt_reduce_defer_chunk hands out chunks of a
// pool instead of one shared scratch (a full pool flushes what is pending and rewinds).  Used by stlt_train_backward: ~55 small
// reduction launches per step become 2 - 3.
struct StltReduceEntry { const float* slabs; float* dst[3]; int64_t stride, n; int n_slabs, n_dst; };
constexpr int STLT_REDUCE_DEFER_MAX = 48;
struct StltReduceDefer {
  StltReduceEntry e[STLT_REDUCE_DEFER_MAX];
  int n = 0;
  hipStream_t s = nullptr;
  float* pool = nullptr;
  size_t pool_floats = 0, used = 0;
  int err = 0;  // first error of a flush made on the way (pool wrap): returned by the next explicit flush
};
void stlt_reduce_defer_set(StltReduceDefer* d);  // nullptr: off
int stlt_reduce_defer_flush(StltReduceDefer* d);
float* stlt_reduce_defer_chunk(StltReduceDefer* d, size_t floats, float* fallback, int* err);
int launch_reduce_slabs3(const float* slabs, int64_t stride, int n_slabs, float* dst0, float* dst1, float* dst2, int64_t n, int accumulate,
                         hipStream_t s);  // three destinations of n columns each, side by side in the slabs
int launch_attn(const float* qkv, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, int64_t dh,
                float* ctx, int kid, hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0);
int launch_attn16(const float* qkv, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, float* ctx, int reverse, hipStream_t s,
                  bool* taken, StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0);  // attn16.hip; *taken = false: not this kernel's shape
// attn_bwdx16.hip: the same for queries and keys from different buffers (cross-attention), no causal mask, at most 48 tokens on either side
int launch_attn_bwdx16(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* dctx, const uint8_t* kpm,
                       int64_t S, int64_t Lq, int64_t Lk, int64_t H, float* dq, int64_t lddq, float* dk, float* dv, int64_t lddkv, StltDrop dr,
                       uint32_t site, hipStream_t s, bool* taken);
int launch_attn_bwd16(const float* qkv, const float* dctx, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, float* dqkv,
                      StltDrop dr, uint32_t site, float* scratch, int want_colsum, int* chunks_out, hipStream_t s,
                      bool* taken);  // attn_bwd16.hip; *taken = false: not this kernel's shape
// mhsa.hip: in-projection + attention core in one kernel (sequences of <= 64 tokens, 64-channel heads); qkv_out: also write the
// packed projections (training tape); dr / site: dropout of the probabilities (needs qkv_out)
bool stlt_mhsa_fused_takes(int64_t L, int64_t H, int64_t d, int causal);           // the kernel has this shape
bool stlt_mhsa_fused_pays(int64_t S, int64_t L, int64_t H, int64_t d, int causal);  // ... and is the faster form for S sequences (whole-path dispatch)
bool stlt_fused_mhsa_on(int causal);  // api.hip: the whole-path entry points use the fused kernel (STLT_FUSED_MHSA=0 / STLT_FUSED_MHSA_SPATIAL switch it)
int launch_mhsa_fused(const float* x, const float* w_in, const float* b_in, const uint8_t* kpm, int64_t S, int64_t L, int64_t H,
                      int64_t d, float* ctx, hipStream_t s, int causal = 1, float* qkv_out = nullptr,
                      StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0);
struct AttnBwdRagged;
int launch_attn_bwd16_ragged(const float* qkv, const float* dctx, const AttnBwdRagged& rg, int causal, int64_t H, float* dqkv, StltDrop dr,
                             uint32_t site, float* scratch, int want_colsum, int* chunks_out, hipStream_t s, bool* taken);  // attn_bwd16.hip, ragged layout
int launch_attn_general(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kpm,
                        int causal, int64_t S, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float* ctx, int kid,
                        hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0);
int launch_add_layernorm(const float* x, int64_t ldx, const float* res, int64_t ldres, const float* w, const float* b,
                         float eps, int64_t M, int64_t d, float* out, int64_t ldout, hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull},
                         uint32_t site = 0, const int* drop_rows = nullptr);  // drop_rows: row -> row index used for the dropout mask
int launch_frames_embed(const float* spatial, int64_t row_stride, const int64_t* frame_types, const float* pos_table,
                        const float* type_table, const float* ln_w, const float* ln_b, float eps, int64_t B, int64_t T,
                        int64_t d, float* out, hipStream_t s, float* pre_out = nullptr, StltDrop dr = StltDrop{0u, 1.0f, 0ull},
                        const int* src_index = nullptr, int64_t n_rows = 0);
int launch_gather_last(const float* x, const int64_t* lengths, int64_t B, int64_t T, int64_t d, float* out,
                       hipStream_t s);

// ragged layout (ragged.hip): index of the real tokens / frames of a padded batch, see RaggedIndex
struct RaggedIndex {
  int *t_seg_start, *t_seg_end, *t_orig;              // per compacted token row: its frame's first row, one past its last row; token index in the padded batch
  int *f_seg_start, *f_seg_end, *f_orig, *f_cls_row;  // per compacted frame row: its clip's first / past-last frame row; frame index b*T+t; token row of its CLS object
  int *last_row;                                      // per clip: compacted frame row of frame lengths[b]-1
  int *clip_tok, *clip_frm, *clip_tok_off, *clip_frm_off;  // per clip: real tokens / frames and their exclusive prefix sums (clip_frm_off has B+1 entries)
  int *f_row_of;                                      // per padded frame b*T+t: its compacted frame row, -1 if padded
  int *sp_grp_ptr;                                    // training: token-row offsets of groups of `frames_per_group` whole frames (see launch_ragged_groups)
  int *counts;                                        // [0] real tokens, [1] real frames, [2] != 0: input breaks the collater contract
};
// sp_grp_ptr[g] = first token row of compacted frame g*frames_per_group (g = 0..n_groups), the last entry = n_tokens
// padded layout: ix.f_cls_row[f] = f*N, ix.last_row[b] = b*T + lengths[b]-1
int launch_padded_rows(const int64_t* lengths, int64_t B, int64_t T, int64_t N, const RaggedIndex& idx, hipStream_t s);
int launch_ragged_groups(const RaggedIndex& idx, int64_t n_tokens, int64_t n_frames, int frames_per_group, hipStream_t s);
int launch_scatter_rows(const float* src, const int* rows, int64_t n, int64_t d, float* dst, int64_t dst_rows, hipStream_t s, const int* n_dev = nullptr);  // dst = 0; dst[rows[i]] = src[i] (i < *n_dev when given)
size_t ragged_index_bytes(int64_t B, int64_t T, int64_t N);
RaggedIndex ragged_index_carve(void* base, int64_t B, int64_t T, int64_t N);
int launch_ragged_index(const uint8_t* kpm_boxes, const uint8_t* kpm_frames, const int64_t* lengths, int64_t B, int64_t T,
                        int64_t N, const RaggedIndex& idx, hipStream_t s);
int launch_gather_rows(const float* src, int64_t ld, const int* rows, int64_t n, int64_t d, float* out, hipStream_t s);
// the caller's row counts instead of a read-back: index entries up to the caller's counts made safe, and the result poisoned (NaN) when they are not the index's
int launch_ragged_host_counts(const RaggedIndex& idx, int64_t n_tok, int64_t n_frm, int64_t max_tok, int64_t max_frm, hipStream_t s);
int launch_ragged_poison(const RaggedIndex& idx, int64_t n_tok, int64_t n_frm, bool allow_more, float* out, int64_t n, hipStream_t s);
// hidden size / head count a model or block may have: any head dim up to 256 (64: the MFMA kernels; others: attn_any.hip); rows are
// moved 16 bytes at a time (hidden sizes that are not multiples of 32 run their products on gemm_any.hip)
inline bool stlt_heads_ok(int64_t d, int64_t H) { return d > 0 && H > 0 && d % H == 0 && d / H <= 256 && d % 4 == 0; }
// attn_any.hip: the same attention for head dims other than 64 (vector ALU; forward <= 1024 keys, backward <= 256 tokens a side)
int launch_attn_any_fwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kpm, const int* seg_start,
                        const int* seg_end, int causal, int64_t n_q, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float* ctx, hipStream_t s,
                        StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0);
int launch_attn_any_bwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* dctx, const uint8_t* kpm,
                        const int* grp_ptr, const int* seg_start, const int* seg_end, int max_rows, int causal, int64_t n_groups, int64_t Lq,
                        int64_t Lk, int64_t H, int64_t dh, float* dq, int64_t lddq, float* dk, float* dv, int64_t lddkv, hipStream_t s,
                        StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0);
int launch_attn_ragged(const float* qkv, const int* seg_start, const int* seg_end, int causal, int64_t M, int64_t H, int64_t dh,
                       float* ctx, int kid, hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0);

// backward kernels (backward.hip)
int64_t ln_bwd_scratch_floats(int64_t d);
int launch_ln_bwd(const float* dy, int64_t lddy, const float* a, int64_t lda, const float* b2, int64_t ldb, const float* w,
                  float eps, int64_t M, int64_t d, float* ds, int64_t ldds, float* g_w, float* g_b, float* scratch,
                  hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site_b2 = 0, float* ds_drop = nullptr,
                  uint32_t site_dy = 0, float* g_colsum = nullptr,  // g_colsum += column sums of the branch gradient (ds_drop, else ds)
                  const int* drop_rows = nullptr);
int launch_colsum_acc(const float* x, int64_t ld, int64_t M, int64_t N, float* g, float* scratch, hipStream_t s);
int launch_gelu_fwd(const float* u, float* h, int64_t n, hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0,
                    const int* drop_rows = nullptr, int64_t ncols = 0);
int launch_gelu_bwd(const float* dh, const float* u, float* du, int64_t n, hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull},
                    uint32_t site = 0, const int* drop_rows = nullptr, int64_t ncols = 0);
int stlt_ffn_hidden_backward_fused(const float* df, const float* lin2_w, const float* u, float* du, int64_t rows, int64_t d, float* g_lin1_b,
                                   float* cs_part, StltDrop dr, uint32_t site, hipStream_t s, bool* taken);  // train.hip
int launch_gelu_bwd_colsum(const float* dh, const float* u, float* du, int64_t M, int64_t N, float* g_colsum, float* scratch,
                           hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0, const int* drop_rows = nullptr);  // scratch >= 512*N floats
// ragged attention backward: groups of whole segments (rows [grp_ptr[g], grp_ptr[g+1]), at most max_rows <= 64 each)
struct AttnBwdRagged { const int* grp_ptr; const int* seg_start; const int* seg_end; int64_t n_groups; int64_t n_rows; int max_rows; };
int launch_attn_bwd(const float* qkv, const float* dctx, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H,
                    int64_t dh, float* dqkv, hipStream_t s, StltDrop dr = StltDrop{0u, 1.0f, 0ull}, uint32_t site = 0,
                    float* g_colsum = nullptr, float* scratch = nullptr, const AttnBwdRagged* ragged = nullptr);  // ragged: L = longest possible segment  // g_colsum (3*H*dh) += column sums of dqkv; scratch >= 256*3*H*dh floats
int64_t embed_bwd_scratch_floats(int64_t n_tokens, int64_t C, int64_t d);
int launch_embed_bwd(const float* dx, const int64_t* categories, const float* boxes, const float* scores, int64_t C,
                     int64_t n_tokens, int64_t d, float* g_cat, float* g_box_w, float* g_box_b, float* g_score_w,
                     float* g_score_b, float* scratch, hipStream_t s, const int* src_index = nullptr);
int launch_frames_bwd(const float* ds, const int64_t* frame_types, int64_t B, int64_t T, int64_t N, int64_t d,
                      float* dx_spatial, float* g_pos, float* g_type, float* scratch, hipStream_t s,
                      const int* row_of = nullptr);  // scratch >= 16*(T+5)*d floats
int launch_scatter_last(const float* dh, const int64_t* lengths, int64_t B, int64_t T, int64_t d, float* dout, hipStream_t s);
int launch_small_gemm(const float* a, int64_t sam, int64_t sak, const float* b, int64_t sbk, int64_t sbn, float* c,
                      int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, hipStream_t s);
