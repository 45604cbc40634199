// The training-loop context behind include/stlt_hip.h's opaque `stlt_ctx` handle: everything the library RETAINS between calls on behalf of
// one training loop lives here and nowhere else —
//   * the set of transposed weight copies of a step (wt_cache.hip: stlt_ctx_wt_refresh .. stlt_ctx_wt_clear),
//   * the queue of deferred block weight gradients (blocks.hip: stlt_ctx_dw_defer / _flush),
//   * the side stream + events of the reverse sweep's weight-gradient products, one set per device (train.hip).
// A public `*_bwd` / `*_backward` call names its context in its argument list (NULL: none of the three) and makes it current on the calling
// thread for the duration of the call (StltCtxScope — the same idiom as StltGemmScratch); the launchers deep inside read stlt_ctx_current().
// torch's autograd engine runs backward nodes on its own threads: the handle travels in every call, so no thread-affinity is assumed.
#pragma once
#include <map>
#include <mutex>
#include "common.h"

struct StltWtEnt { const float* w; const float* wt; int64_t n_out, k_in; };
constexpr int STLT_DW_DEFER_CAP = 512;
// `busy`: a device's side stream and its four events are one set per context, so a sweep owns them from its first fork to its join — a second
// host thread's sweep with the same context on the same device waits instead of re-recording an event the first one is about to wait on.
struct StltSideDevice { hipStream_t s = nullptr; hipEvent_t ev[4] = {}; bool tried = false, ok = false; std::mutex busy; };
constexpr uint64_t STLT_CTX_MAGIC = 0x53544c5443545831ull;  // "STLTCTX1"

struct stlt_ctx {
  uint64_t magic = STLT_CTX_MAGIC;
  std::mutex mu;  // guards wt*, dw*
  // transposed weight copies: keyed by the weight's first byte; current from stlt_ctx_wt_refresh to stlt_ctx_wt_clear, on `wt_device` only
  std::map<uintptr_t, StltWtEnt> wt;
  int wt_device = -1;
  hipEvent_t wt_ready = nullptr;      // recorded behind the transposes on the refresh stream
  hipStream_t wt_stream = nullptr;    // the refresh stream (consumers on it are ordered already)
  hipStream_t wt_waited[4] = {};      // consumer streams that have been made to wait for this refresh
  int wt_n_waited = 0;
  bool wt_wait_all = false;           // more than four consumer streams: wait on every call
  long long wt_hits = 0;              // input-gradient products launched on a copy
  // deferred block weight gradients
  StltWeightGradItem dw[STLT_DW_DEFER_CAP];
  int dw_n = 0;
  bool dw_on = false;
  StltSideDevice side[STLT_MAX_DEVICES];
};

inline bool stlt_ctx_valid(const stlt_ctx* c) { return c && c->magic == STLT_CTX_MAGIC; }
