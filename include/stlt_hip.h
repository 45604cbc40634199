/*
 * stlt_hip.h — C-ABI of libstlt_hip.so: the MI355X (gfx950) STLT forward hot path.
 *
 * The reference (gorjanradevski/revisiting-spatial-temporal-layouts) has no FFI: its boundary is the
 * Python class surface of src/modelling/models.py.  This header is the inner boundary the drop-in
 * `Stlt` / `StltBackbone` modules bind through ctypes; every entry point names the reference code it
 * replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - extern "C"; every function returns 0 on success, a negative STLT_E* code on argument errors,
 *     or a positive hipError_t; stlt_last_error() returns a thread-local message.
 *   - all tensor arguments are raw DEVICE pointers owned by the caller (PyTorch allocations); the
 *     library never allocates or frees device memory.  It retains caller pointers in ONE place only: a training-loop context
 *     (stlt_ctx, below), which the caller creates, names in the calls that may use it, and destroys.  fp32, row-major,
 *     contiguous unless a leading dimension (ld*) is given in ELEMENTS.  ids/lengths int64, masks uint8 (1 = padded / masked).
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no implicit synchronisation — the one exception
 *     is STLT_FLAG_SKIP_PADDING without the caller's row counts (stlt_inputs.n_real_tokens / n_real_frames), which reads
 *     two row counts back (one stream synchronisation per call); everything else is a fixed launch sequence that can be
 *     captured in a hipGraph.
 *   - per-thread state: the error string and the scratch lent with stlt_gemm_set_scratch (both thread-local).
 *     Process-wide: write-once caches of device properties and the routing switches (stlt_set_gemm_small_tiles,
 *     stlt_set_gemm_split_bf16, stlt_set_train_side_stream: plain integers, set them before the calls they steer, not
 *     concurrently with them).  Nothing else is process-wide: what a training loop leaves in the library between calls —
 *     the transposed weight copies of its step, its queue of deferred block weight gradients, the side stream + events of
 *     its reverse sweeps — lives in the stlt_ctx it passes to the `*_bwd` / `*_backward` calls (NULL: none of the three).
 *     A context is internally locked: calls naming the same context from several host threads are safe (reverse sweeps on
 *     the same device are serialised from their first fork to their join); different contexts never interact, so two
 *     training loops in one process — or a plain autograd backward of another model in the middle of a trainer's step —
 *     cannot read each other's copies or queue into each other's flush.
 *   - plain C: this header compiles as C99 and as C++ (tests/test_host_cpu.py builds a C client against the library).
 */
#ifndef STLT_HIP_H
#define STLT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STLT_VERSION 110  /* 110: training-loop contexts (stlt_ctx) in the backward calls; block backward buffers split into keep / work */

#define STLT_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define STLT_EWORKSPACE (-2) /* workspace too small */

#define STLT_ACT_NONE 0
#define STLT_ACT_GELU 1  /* exact erf GELU */
#define STLT_ACT_RELU 2  /* nn.TransformerEncoderLayer default activation (appearance branch, models.py:239-246) */

typedef void* stlt_stream_t; /* hipStream_t */

int stlt_version(void);
const char* stlt_last_error(void);

/* ---- training-loop context -------------------------------------------------------------------------------------------
 * The reference's train() (src/train.py:102-135) is one loop over one model; its state between statements lives in torch objects.
 * Here the three things a loop leaves INSIDE the library between calls hang off an explicit handle:
 *   transposed weight copies   stlt_ctx_wt_refresh .. stlt_ctx_wt_clear: current for the calls that name the context, on the device of
 *                              the refresh; every such call's stream is ordered behind the transposes by the library (an event
 *                              recorded at the end of the refresh), whatever stream the refresh ran on
 *   deferred weight gradients  stlt_ctx_dw_defer / _pending / _flush: queued by the block backwards that name the context
 *   side stream + events       of stlt_train_backward's weight-gradient products: one set per context and device, created by the first
 *                              sweep that wants it (stlt_set_train_side_stream), destroyed with the context
 * stlt_ctx_destroy: the caller has synchronised with what it enqueued through the context; queued products are dropped, copies
 * withdrawn.  Functions taking a context return STLT_EINVAL for a handle that is not live. */
typedef struct stlt_ctx stlt_ctx;
int stlt_ctx_create(stlt_ctx** out);
int stlt_ctx_destroy(stlt_ctx* ctx);

/* K1 — CategoryBoxEmbeddings.forward, src/modelling/models.py:29-39.
 * out[t,:] = LN_eps( cat_table[categories[t]] + boxes[t,0:4]·box_w^T + box_b (+ scores[t]*score_w[:,0] + score_b) )
 * scores may be NULL (key absent from the batch, models.py:33).  d % 4 == 0, d <= 2048. */
int stlt_embed_fwd(const int64_t* categories, const float* boxes, const float* scores,
                   const float* cat_table, int64_t n_categories,
                   const float* box_w, const float* box_b, const float* score_w, const float* score_b,
                   const float* ln_w, const float* ln_b, float eps,
                   int64_t n_tokens, int64_t d, float* out, stlt_stream_t stream);

/* K2/K4/K5/K6/K8 — nn.Linear (F.linear inside F.multi_head_attention_forward, linear1/linear2 of
 * nn.TransformerEncoderLayer as configured at models.py:46-52,118-124; fc1/fc2 models.py:158-163).
 * y[m, n] = act( sum_k x[m*ldx + k] * w[n*K + k] + bias[n] ),  w is (N,K) row-major (torch (out,in)).
 * M, N arbitrary.  K % 32 == 0 (every hidden size the released checkpoints use): f32-input MFMA (v_mfma_f32_32x32x2_f32), fp32
 * accumulate.  Any other K (hidden sizes like 100 or 200, which configs.py:92-111 allows): the same product and epilogues on
 * csrc/gemm_any.hip (zero-filled tiles staged through LDS by ordinary loads) — a compatibility path with the same tolerances, not a
 * tuned one. */
int stlt_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias,
                    float* y, int64_t ldy, int64_t M, int64_t N, int64_t K, int act, stlt_stream_t stream);

/* General product on the same kernel, used by the backward pass of nn.Linear (autograd of F.linear in the reference):
 *   c (M,N) = opA(a)·opB(b) [+ r]   with contraction length K (a multiple of 32 for the tuned kernel; any other K runs on the
 *                                   fallback of csrc/gemm_any.hip, n_split = 1 only)
 *   transA=0: a is (M,K) row-major, lda;  transA=1: a is (K,M) row-major, lda   (dW = dY^T·X)
 *   transB=0: b is (N,K) row-major, ldb;  transB=1: b is (K,N) row-major, ldb   (dX = dY·W)
 * r (nullable, ldr) is added in the epilogue (residual gradient).  n_split > 1 splits the contraction: split s writes
 * its partial product to c + s*slab_stride; sum them with stlt_reduce_slabs (deterministic, no atomics).
 * Rows of a contraction-major operand beyond the logical K must be zero-filled by the caller (K is rounded up). */
int stlt_gemm(int transA, int transB, const float* a, int64_t lda, const float* b, int64_t ldb,
              const float* r, int64_t ldr, float* c, int64_t ldc, int64_t slab_stride,
              int64_t M, int64_t N, int64_t K, int n_split, stlt_stream_t stream);
/* The same nn.Linear forward (plus an optional add-source r: y = act(x·Wᵀ + b) + r, r only with STLT_ACT_NONE) on the small-tile
 * kernel (csrc/gemm16.hip): whole tiles, no stream-K, no fix-up launch.  The `tile` parameter is columns | rows << 16 with rows 0 = 128:
 * 128 rows x {48, 64, 96, 128, 144, 192} columns, 64 rows x {64, 96, 128, 160, 192, 256}, 32 rows x {128, 192, 256}.  stlt_linear_fwd and every whole-path / training / block entry point route a product here by themselves when its launch
 * would be under-filled on the 256 x 128 tiles and the small tiles are estimated faster (few rows: the temporal tower at the
 * reference's default batch of 64 clips, the fusion models' 2048 / 2112-row blocks); stlt_linear_small_choice returns the tile
 * (same encoding) that dispatch picks for (M, N, K) — 0: the product stays on the large tiles; STLT_GEMM16=0 in the environment disables the routing.
 * This entry point runs the kernel on any shape it can take (K % 32 == 0, K >= 64, N % 4 == 0, pitches % 4 == 0) with the tile
 * given (tests, A/B measurements); other shapes return STLT_EINVAL.  Same result as stlt_linear_fwd to fp32 rounding. */
int stlt_linear_small_fwd(const float* x, int64_t ldx, const float* w, const float* bias, const float* r, int64_t ldr, float* y, int64_t ldy,
                          int64_t M, int64_t N, int64_t K, int act, int tile, stlt_stream_t stream);
int stlt_linear_small_choice(int64_t M, int64_t N, int64_t K);
/* The input gradient of that Linear on the same kernel: dx (M, k_in; ld_dx) = dy (M, n_out; ld_dy) · w (n_out, k_in) (+ r), the weight
 * read as it lies (no transposed copy: the [k][n] image is gathered inside the kernel).  The training sweeps (stlt_train_backward,
 * the block backwards, stlt_linear_bwd) route their under-filled dX products here by the same launch-time estimate.  tile != 0 runs it
 * on that tile (columns | rows << 16), always reading w as it lies; tile == 0 routes by the estimate — STLT_EINVAL when the estimate
 * leaves the product to the large tiles (stlt_input_grad_small_choice tells beforehand) — and, when `ctx` holds a current transposed
 * copy of w (stlt_ctx_wt_refresh), runs the product as a forward product on the copy.  n_out % 32 == 0, n_out >= 64, k_in % 4 == 0.
 * Same result as stlt_gemm(0, 1, ...) to rounding.  ctx may be NULL. */
int stlt_input_grad_small(const float* dy, int64_t ld_dy, const float* w, int64_t n_out, int64_t k_in, const float* r, int64_t ldr, float* dx,
                          int64_t ld_dx, int64_t M, int tile, stlt_ctx* ctx, stlt_stream_t stream);
int stlt_input_grad_small_choice(int64_t M, int64_t n_out, int64_t k_in);  /* the tile the routing picks for that input gradient, reading w as it lies: columns | rows << 16 (0: large tiles) */
/* Process-wide routing switch: -1 = by the launch-time estimate (default; STLT_GEMM16 in the environment is the initial value),
 * 0 = every product on the large tiles, 1 = every product the small-tile kernel can take on it (A/B measurements), 128 / 64 / 32 = by
 * the estimate over tiles of that height only (tests, A/B; STLT_GEMM16_ROWS is the environment's form), -2 = back to the initial value
 * (what a test or tool that switched it should leave behind). */
int stlt_set_gemm_small_tiles(int mode);

/* Optional scratch for the calling thread's stlt_linear_fwd / stlt_gemm launches (torch's nn.Linear has no
 * counterpart: this is launch policy).  With scratch lent, a launch whose 256x128 output tiles would leave compute
 * units idle (fewer tiles than CUs, or a ragged last round) is cut into equal contiguous k-step ranges instead
 * ("stream-K"); tiles computed by more than one workgroup are summed in workgroup order by a second kernel, so results
 * stay deterministic.  The buffer must hold stlt_gemm_scratch_bytes() and may only be reused by work ordered after
 * the launch on its stream.  Pass NULL to withdraw.  The whole-path entry points lend a slice of their workspace. */
size_t stlt_gemm_scratch_bytes(void);
int stlt_gemm_set_scratch(void* scratch, size_t bytes);
/* Weight gradients of several nn.Linear modules in ONE launch (what autograd computes product by product for
 * `loss.backward()`, src/train.py:125-127): g_w_i (n_out_i, k_in_i) += dy_i[:rows_i]^T · x_i[:rows_i] for every item.
 * dy_i (rows_i, n_out_i) and x_i (rows_i, k_in_i) row-major; rows_i a multiple of 32 (rows beyond the logical count
 * must be zero in dy or x); n_out_i, k_in_i multiples of 4; 1..32 items; items with g_w == NULL are skipped.  Needs lent
 * stream-K scratch (stlt_gemm_set_scratch).  One persistent stream-K launch over the concatenated k-step space of the
 * items + one fix-up: deterministic (fixed summation order). */
typedef struct { const float* dy; int64_t n_out; const float* x; int64_t k_in; int64_t rows; float* g_w; } stlt_wgrad_item;
int stlt_weight_grad_group(const stlt_wgrad_item* items, int n_items, stlt_stream_t stream);
/* dst[i] = (accumulate ? dst[i] : 0) + sum_s slabs[s*stride + i], i < n */
int stlt_reduce_slabs(const float* slabs, int64_t stride, int n_slabs, float* dst, int64_t n, int accumulate,
                      stlt_stream_t stream);

/* K3 — attention core of F.multi_head_attention_forward as reached from models.py:68-71 (spatial,
 * key-padding mask) and models.py:146-150 (temporal, causal mask of utils/model_utils.py:4-7 + key padding).
 * qkv: (S*L, 3*H*dh) packed rows [q;k;v]; ctx: (S*L, H*dh).  kpm: (S*L) bytes, 1 = key masked.
 * ctx[s,i,h,:] = softmax_j( q_i·k_j/sqrt(dh) + M_ij ) v_j with M_ij = -inf if kpm[s,j] or (causal and j>i).
 * Rows whose keys are all masked produce zeros.  1 <= dh <= 256: dh == 64 (every released checkpoint) runs on the MFMA kernels; any
 * other head dim the reference's configs.py:92-111 allows runs on the vector-ALU kernels of csrc/attn_any.hip (at most 1024 keys per
 * sequence in the forward, 256 tokens a side in the backward) — the same arithmetic, masks, dropout indices and tolerances. */
int stlt_attn_core_fwd(const float* qkv, const uint8_t* kpm, int causal,
                       int64_t S, int64_t L, int64_t H, int64_t dh, float* ctx, stlt_stream_t stream);

/* Fused multi-head self-attention (the north star's "fused MHSA"): the in-projection of nn.MultiheadAttention
 * (models.py:46-52,118-124: in_proj_weight (3d,d) rows [q;k;v], in_proj_bias) and the attention core in ONE kernel:
 * x (S*L, d) -> ctx (S*L, d), the packed QKV tensor never goes to memory.  stlt_mhsa_fused_fwd is the temporal tower's form
 * (causal mask of utils/model_utils.py:4-7 + src_key_padding_mask_frames, models.py:142-150).  Sequences of 1 <= L <= 64 tokens
 * (the reference's layouts are T = layout_num_frames + 1 = 17 / 33, datasets.py:97-113; Action Genome 64) and 64-channel heads
 * (d == 64*H); other shapes return STLT_EINVAL (callers use stlt_linear_fwd + stlt_attn_core_fwd).  Same result as that pair to
 * fp32 rounding. */
int stlt_mhsa_fused_fwd(const float* x, const float* in_proj_w, const float* in_proj_b, const uint8_t* kpm, int64_t S, int64_t L,
                        int64_t H, int64_t d, float* ctx, stlt_stream_t stream);
/* The same kernel with every option: causal = 0 is the spatial tower's form (key-padding mask only, models.py:68-71; L up to
 * ~36 tokens); qkv_out != NULL also writes the packed projections (S*L, 3d) — what a training forward keeps for the reverse
 * sweep; dropout_p > 0 (needs qkv_out) drops attention probabilities with the library's counter mask at `site`, element index
 * ((query token * H + head) << 8) | key position, exactly as stlt_attn_core_fwd_train does, so stlt_attn_core_bwd regenerates it. */
int stlt_mhsa_fused_fwd_ex(const float* x, const float* in_proj_w, const float* in_proj_b, const uint8_t* kpm, int causal, int64_t S,
                           int64_t L, int64_t H, int64_t d, float dropout_p, uint64_t seed, uint32_t site, float* ctx, float* qkv_out,
                           stlt_stream_t stream);
/* 1 when stlt_forward / stlt_backbone_forward / stlt_train_forward run their temporal layers through the fused kernel for this
 * shape at 1024 clips per launch (64-channel heads, T <= 64 frames; the launch-time estimate of csrc/mhsa.hip compares the fused
 * launch with the in-projection + attention-core pair: T = 17, 32, 64 take the fused kernel, T = 33 — 99 of a work item's 128 rows —
 * keeps the pair; STLT_FUSED_MHSA=0 in the environment switches the fused kernel off).  stlt_fused_mhsa_used answers for one
 * launch of S sequences of L tokens (causal: temporal tower, else the spatial tower): small launches (64 clips) go to the pair,
 * whose in-projection then runs on the small-tile kernel. */
int stlt_fused_mhsa_active(int64_t T, int64_t d, int64_t H);
int stlt_fused_mhsa_used(int64_t S, int64_t L, int64_t d, int64_t H, int causal);
/* Cross-attention core (CrossAttentionLayer of CAF/CACNF, models.py:362-382; also self-attention on unpacked buffers):
 * queries q (S*Lq rows, stride ldq floats) attend to keys k / values v (S*Lk rows, stride ldkv).  kpm: (S*Lk) bytes over
 * the KEY tokens (pass zeros for no padding mask).  ctx: (S*Lq, H*dh).  causal requires Lq == Lk. */
int stlt_attn_cross_fwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kpm,
                        int causal, int64_t S, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float* ctx,
                        stlt_stream_t stream);

/* K3 on a ragged layout: self-attention over M compacted rows of a packed (M, 3*H*dh) buffer cut into variable-length
 * segments (one frame's objects, one clip's frames) — what STLT_FLAG_SKIP_PADDING runs instead of the padded K3.
 * seg_start[r] / seg_end[r] (int32, device): first row / one past the last row of row r's segment; segments are
 * contiguous and cover [0, M).  ctx[r] = softmax over the keys of r's segment (only those at rows <= r when causal). */
int stlt_attn_ragged_fwd(const float* qkv, const int32_t* seg_start, const int32_t* seg_end, int causal, int64_t M,
                         int64_t H, int64_t dh, float* ctx, stlt_stream_t stream);

/* Residual + LayerNorm (norm1/norm2 of nn.TransformerEncoderLayer, eps 1e-5; ClassificationHead.layer_norm
 * models.py:159,163 with res == NULL).  out[m,:] = LN_eps( x[m*ldx + :] + res[m*ldres + :] ). */
int stlt_add_layernorm_fwd(const float* x, int64_t ldx, const float* res, int64_t ldres,
                           const float* ln_w, const float* ln_b, float eps,
                           int64_t M, int64_t d, float* out, int64_t ldout, stlt_stream_t stream);

/* K7 — CLS select (models.py:79) + FramesEmbeddings.forward (models.py:98-111).
 * out[b,t,:] = LN_eps( spatial[(b*T+t)*row_stride + :] + pos_table[t] + type_table[frame_types[b,t]] ). */
int stlt_frames_embed_fwd(const float* spatial, int64_t row_stride, const int64_t* frame_types,
                          const float* pos_table, const float* type_table,
                          const float* ln_w, const float* ln_b, float eps,
                          int64_t B, int64_t T, int64_t d, float* out, stlt_stream_t stream);

/* K8a — Stlt.forward gather, models.py:189-192: out[b,:] = x[b, lengths[b]-1, :] for batch-major x (B,T,d). */
int stlt_gather_last_fwd(const float* x, const int64_t* lengths, int64_t B, int64_t T, int64_t d,
                         float* out, stlt_stream_t stream);

/* Collater on the device — padding + mask half of StltCollater.__call__, src/modelling/datasets.py:243-288 (pad_sequence:
 * src/utils/data_utils.py:93-102).  Inputs are the per-video tensors of StltDataset.__getitem__ (datasets.py:52-125)
 * concatenated along the frame axis: video b owns frames [frame_offsets[b], frame_offsets[b+1]); outputs are the padded
 * (B,T,N,.) batch and both key-padding masks (uint8, 1 = padded).  Frames past a video's length carry the CLS object in
 * slot 0 (category cls_id, box [0,0,1,1], score 1) and frame type 0.  scores_ragged/scores are NULL together
 * (datasets.py:253-260 keeps scores only for action_genome). */
int stlt_collate_fwd(const int64_t* categories_ragged, const float* boxes_ragged, const float* scores_ragged,
                     const int64_t* frame_types_ragged, const int64_t* frame_offsets, int64_t B, int64_t T, int64_t N,
                     int64_t cls_id, int64_t* categories, float* boxes, float* scores, int64_t* frame_types,
                     uint8_t* kpm_boxes, uint8_t* kpm_frames, stlt_stream_t stream);

/* ---- whole-path entry points (host-side orchestration in native code) ---- */

typedef struct {
  const float *in_proj_w, *in_proj_b;   /* (3d,d) rows [q;k;v], (3d) */
  const float *out_proj_w, *out_proj_b; /* (d,d), (d) */
  const float *lin1_w, *lin1_b;         /* (4d,d), (4d) */
  const float *lin2_w, *lin2_b;         /* (d,4d), (d) */
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b; /* (d) each, eps 1e-5 */
} stlt_layer_params;

typedef struct {
  int64_t d, H, n_categories, n_spatial, n_temporal, n_classes, n_positions;
  float ln_eps; /* config.layer_norm_eps (1e-12) — the three explicit LayerNorms only */
  const float *cat_emb, *box_w, *box_b, *score_w, *score_b, *emb_ln_w, *emb_ln_b; /* models.py:19-27 */
  const float *pos_emb, *type_emb, *frames_ln_w, *frames_ln_b;                    /* models.py:88-93 */
  const stlt_layer_params* spatial;  /* HOST array [n_spatial]  (models.py:53-55) */
  const stlt_layer_params* temporal; /* HOST array [n_temporal] (models.py:126-128) */
  const float *fc1_w, *fc1_b, *head_ln_w, *head_ln_b, *fc2_w, *fc2_b; /* models.py:158-160; may be NULL for backbone-only */
} stlt_params;

typedef struct {
  int64_t B, T, N;
  const int64_t* categories;   /* (B,T,N) */
  const float* boxes;          /* (B,T,N,4) */
  const float* scores;         /* (B,T,N) or NULL */
  const uint8_t* kpm_boxes;    /* (B,T,N)  src_key_padding_mask_boxes */
  const int64_t* frame_types;  /* (B,T) */
  const uint8_t* kpm_frames;   /* (B,T)    src_key_padding_mask_frames */
  const int64_t* lengths;      /* (B) */
  /* STLT_FLAG_SKIP_PADDING only, optional (0 = unknown): the batch's real rows as the collater can count them on the host —
   * n_real_frames = number of zeros in kpm_frames, n_real_tokens = number of zeros in kpm_boxes inside those frames.  With both given the
   * call reads nothing back: no stream synchronisation, capturable in a hipGraph.  In the inference calls (stlt_forward, stlt_backbone_forward,
   * stlt_caf_forward_flags) the two numbers may be UPPER BOUNDS: the rows between the real counts and the bounds are computed as
   * self-contained dummy rows nobody reads, so one captured graph serves every batch whose counts stay below the bounds it was captured
   * with (a bucket, a percentile of the dataset).  The training calls need the exact counts (a dummy row would leave a gradient).  Either way
   * they are verified on the device: real counts above the bounds (training: different from them), or masks that break the collater contract,
   * give NaN results instead of an error return. */
  int64_t n_real_tokens, n_real_frames;
} stlt_inputs;

#define STLT_FLAG_CLS_ONLY_LAST_SPATIAL 1 /* last spatial layer: Q/out-proj/FFN on the CLS rows only (the only rows read, models.py:79) */
#define STLT_FLAG_SKIP_PADDING 4 /* stlt_forward with out_btd == NULL: compute the real (unmasked) tokens and frames only.  Same logits: a padded row is
                                   masked as a key everywhere and never read as a query result.  Implies both flags above.  Needs collater-shaped
                                   masks (slot 0 of a real frame unmasked, frame lengths-1 real); synchronises the stream once per call unless stlt_inputs carries
                                   the batch's two row counts (n_real_tokens / n_real_frames). */
#define STLT_FLAG_TRAIN_UPPER_ONLY 8   /* stlt_train_backward: stop after the prediction head and the temporal tower (their gradients are final) */
#define STLT_FLAG_TRAIN_LOWER_ONLY 16  /* stlt_train_backward: resume from there (frames embeddings, spatial tower, token embeddings); the scratch must be
                                          untouched in between.  Lets a data-parallel caller all-reduce the upper gradients while the lower half runs. */
#define STLT_FLAG_TRAIN_BACKBONE 32     /* stlt_train_forward / _backward as the StltBackbone of a fusion model (models.py:136-152 inside :446-483): no
                                         * prediction head, every temporal layer on every frame; the forward's `logits` argument receives the (B*T, d)
                                         * backbone output and the backward's `dlogits` argument is its gradient (B*T, d).  Not with STLT_FLAG_SKIP_PADDING. */
#define STLT_FLAG_LAST_ROW_ONLY_TEMPORAL 2 /* stlt_forward with out_btd == NULL: last temporal layer's out-proj/FFN on the rows at lengths-1 only (models.py:189-192) */

/* bytes of scratch the whole-path calls need for this shape */
size_t stlt_workspace_bytes(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_classes);

/* StltBackbone.forward, models.py:136-152.  out_btd is batch-major (B,T,d); the module returns its (T,B,d) view. */
int stlt_backbone_forward(const stlt_params* p, const stlt_inputs* in, void* workspace, size_t workspace_bytes,
                          int flags, float* out_btd, stlt_stream_t stream);

/* Stlt.forward, models.py:185-195: backbone -> gather at lengths-1 -> ClassificationHead.  logits (B,n_classes).
 * out_btd may be NULL (then the backbone output lives in the workspace). */
int stlt_forward(const stlt_params* p, const stlt_inputs* in, void* workspace, size_t workspace_bytes,
                 int flags, float* out_btd, float* logits, stlt_stream_t stream);

/* ---- CAF / CACNF on precomputed appearance features (reference models.py:230-271, 286-298, 328-549) ----
 * The layout branch is the StltBackbone above; the appearance branch starts from the R3D-50 feature map the reference's
 * Resnet3D.forward_features returns, (B, 2048, 2,4,4) = (B, feat_channels, app_tokens) row-major, given by the caller. */
typedef struct {  /* SelfAttentionLayer / CrossAttentionLayer: nn.MultiheadAttention + LayerNorm(eps = ln_eps), models.py:345-382 */
  const float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *ln_w, *ln_b;
} stlt_attn_block_params;
typedef struct {  /* FeedforwardModule, models.py:328-342 */
  const float *lin1_w, *lin1_b, *lin2_w, *lin2_b, *ln_w, *ln_b;
} stlt_ffn_block_params;
typedef struct {  /* CrossModalModule, models.py:385-431; cross_attn is shared by both directions, appearance_ffn IS a self-attention layer */
  stlt_attn_block_params cross_attn, layout_attn, appearance_attn, appearance_ffn;
  stlt_ffn_block_params layout_ffn;
} stlt_crossmodal_params;
typedef struct {  /* ClassificationHead (d->d->K) or FusionHead (2d->d->K): fc1, LayerNorm(ln_eps), fc2 */
  const float *fc1_w, *fc1_b, *ln_w, *ln_b, *fc2_w, *fc2_b;
} stlt_head_params;
typedef struct {
  stlt_params layout;                 /* layout_branch (head fields unused) */
  int64_t feat_channels, app_tokens;  /* 2048, 32 */
  const float *proj_w, *proj_b;       /* projector Conv3d 1x1x1 == Linear (d, feat_channels), models.py:236-238 */
  const float *cls_token, *pos_embed; /* (d), (app_tokens+1, d), models.py:247-250 */
  int64_t n_app_layers;
  const stlt_layer_params* app_layers; /* HOST array; ReLU encoder layers, LN eps 1e-5, models.py:239-246 */
  int64_t n_fusion;
  const stlt_crossmodal_params* fusion; /* HOST array */
  stlt_head_params fusion_head;        /* FusionHead: classifier (CAF) / fusion_classifier (CACNF) */
  stlt_head_params layout_head, appearance_head; /* CACNF only (ClassificationHead x2); fc1_w == NULL for CAF */
} stlt_caf_params;

size_t stlt_caf_workspace_bytes(int64_t B, int64_t T, int64_t N, int64_t d, int64_t feat_channels, int64_t app_tokens,
                                int64_t n_classes);
/* CrossAttentionFusion.forward (models.py:486-498) -> logits_caf (B,K).  With the two extra heads given it is
 * CrossAttentionCentralNetFusion.forward (models.py:520-549): logits_stlt, logits_resnet3d, logits_caf and
 * logits_ensemble = their mean (all (B,K); the three extra outputs may be NULL for CAF). */
int stlt_caf_forward(const stlt_caf_params* p, const stlt_inputs* in, const float* appearance_features, void* workspace,
                     size_t workspace_bytes, float* logits_caf, float* logits_stlt, float* logits_resnet3d,
                     float* logits_ensemble, stlt_stream_t stream);
/* The same with STLT_FLAG_SKIP_PADDING accepted in `flags`: the layout branch runs on the real tokens / frames only (the
 * collater's masks, as for stlt_forward); padded frames' rows of the (B,T,d) layout state are zero — downstream they are
 * only masked keys (models.py:403-431) — so the logits are those of the padded schedule to rounding.  Like stlt_forward
 * with that flag it synchronises the stream once (two row counts are read back to size the launches) unless stlt_inputs carries them. */
int stlt_caf_forward_flags(const stlt_caf_params* p, const stlt_inputs* in, const float* appearance_features, void* workspace,
                           size_t workspace_bytes, int flags, float* logits_caf, float* logits_stlt, float* logits_resnet3d,
                           float* logits_ensemble, stlt_stream_t stream);

/* ---- training step (reference src/train.py:119-135: forward, loss.backward(); optimiser step further below) ----
 * stlt_train_forward runs every layer on every row — except that the last layer of each tower runs its out-proj /
 * norms / FFN only on the rows read afterwards (CLS row per frame, frame lengths-1 per clip; same loss and gradients,
 * dropout masks keep the rows' original indices) — and records every intermediate the reverse sweep needs in `tape`
 * (fp32; rows padded to a multiple of 32; the caller allocates it ZERO-FILLED once and the library never writes the
 * padding).  stlt_train_backward is the reverse sweep: given dlogits (B, n_classes) it ACCUMULATES (+=) parameter
 * gradients into the buffers named by `grads` — the same struct as the parameters, every pointer being the gradient
 * buffer of that parameter or NULL to skip it (frozen / unused parameters; models.py:172-174).  `scratch` must be
 * zero-filled once by the caller as well.  Embedding rows at padding_idx 0 receive no gradient (models.py:22,91).
 * Dropout (nn.Dropout after both embedding LayerNorms; attention probabilities, dropout1, FFN dropout and dropout2 of
 * every encoder layer — SURVEY.md App. B) is a counter-based mask: with key = splitmix64-finaliser(seed*0x9E3779B97F4A7C15
 * + s*0xD1B54A32D192ED03), element idx of site s is kept iff the keyed 32-bit mixer of csrc/common.h (stlt_keep_k: two
 * multiply-xorshift rounds over idx + key_lo, key_hi folded in between) is >= p*2^32; kept values are scaled by 1/(1-p).
 * The backward recomputes the masks from (p, seed): pass the same values to both calls.  p = 0 disables it.
 * Attention backward supports sequences of up to 256 tokens (the position table).  At most STLT_TRAIN_MAX_CATEGORIES object
 * categories (the embedding-gradient kernel keeps per-category sums in LDS; the reference's vocabularies have 4 and 38):
 * stlt_train_scratch_bytes returns 0 and stlt_train_forward / _backward return STLT_EINVAL above it, before anything runs. */
#define STLT_TRAIN_MAX_CATEGORIES 128
/* stlt_train_backward, when it is given a context, runs the weight-gradient products (off the dX chain) on the context's second stream for
 * the device, forked from and joined to the caller's stream with events inside the call (every exit path, error returns included, joins);
 * without a context, or with this switch at 0, every launch stays on the caller's stream (A/B runs, per-kernel event timing without
 * overlap).  The switch is process-wide; STLT_TRAIN_DW_STREAM in the environment is its initial value (default on);
 * on < 0 goes back to that value.  stlt_get_train_side_stream returns the setting in force (1 / 0), so that a caller can restore it. */
int stlt_set_train_side_stream(int on);
int stlt_get_train_side_stream(void);
size_t stlt_train_tape_bytes(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_spatial, int64_t n_temporal);
size_t stlt_train_scratch_bytes(int64_t B, int64_t T, int64_t N, int64_t d, int64_t n_categories);
/* flags: 0, or STLT_FLAG_SKIP_PADDING (the same value in both calls of a step): forward and reverse sweep run over the
 * real tokens / frames only — same loss and gradients when dropout is off; with dropout the masks are drawn per
 * compacted row, i.e. a different (equally valid) random stream than the padded schedule's.  Each call then
 * synchronises the stream once, unless stlt_inputs carries the batch's exact row counts (n_real_tokens / n_real_frames). */
int stlt_train_forward(const stlt_params* p, const stlt_inputs* in, void* tape, size_t tape_bytes, float* logits,
                       float dropout_p, uint64_t dropout_seed, int flags, stlt_stream_t stream);
int stlt_train_backward(const stlt_params* p, const stlt_params* grads, const stlt_inputs* in, const void* tape,
                        size_t tape_bytes, void* scratch, size_t scratch_bytes, const float* dlogits,
                        float dropout_p, uint64_t dropout_seed, int flags, stlt_ctx* ctx, stlt_stream_t stream);

/* ---- optimiser step of the training loop (reference train.py:128-131 with utils/train_inference_utils.py:37-54) ----
 * The reverse sweep writes all parameter gradients into ONE flat fp32 buffer (what a data-parallel run all-reduces).
 * stlt_grad_norm: out[0] = ||g||_2 over the n floats, out[1] = min(1, max_norm / (out[0] + 1e-6)) — the factor
 * torch.nn.utils.clip_grad_norm_ scales the gradients by (max_norm <= 0: out[1] = 1).  scratch: >= 1024 floats.
 * stlt_adamw_step: torch.optim.AdamW.step (decoupled weight decay, bias correction, fp32) over a device table of
 * chunks; element i of a chunk is parameter param[i], its gradient flat_grad[flat_offset + i] (scaled by
 * norm_and_clip[1] when given) and its moments exp_avg / exp_avg_sq at the same flat offset.  step counts from 1. */
typedef struct {
  float* param;
  int64_t flat_offset;
  int32_t n;
  float weight_decay;
} stlt_opt_chunk;
/* ---- per-kernel backward entry points (autograd of the K-row forwards above; the fusion models' training is composed
 * from these, the STLT training step uses the fixed reverse sweep of stlt_train_backward) ----
 * stlt_linear_bwd: y = x·Wᵀ + b (no activation).  dx (M,K) = dy·W (nullable), dw (N,K) += dyᵀ·x (nullable), db (N) +=
 *   column sums of dy (nullable).  scratch: stlt_linear_bwd_scratch_bytes(N).  ctx (nullable): dx may run on the context's transposed copy of w.
 * stlt_attn_bwd: backward of stlt_attn_cross_fwd (and, with q = qkv, k = qkv+d, v = qkv+2d, of stlt_attn_core_fwd):
 *   dq / dk / dv written (not accumulated) with their own leading dimensions; sequences of at most 256 tokens on either
 *   side (above 64 a streamed variant: query tiles of 32, keys / values in tiles through LDS); head dims other than 64: attn_any.hip.
 * stlt_add_layernorm_bwd: out = LN_eps(x + res)·w + b.  ds = gradient wrt the sum (the gradient of both x and res);
 *   g_w / g_b accumulate (nullable).  scratch: stlt_add_layernorm_bwd_scratch_bytes(d).
 * stlt_gelu_fwd / stlt_gelu_bwd: exact-erf GELU and du = dh * gelu'(u), n a multiple of 4. */
size_t stlt_linear_bwd_scratch_bytes(int64_t N);
int stlt_linear_bwd(const float* x, const float* w, const float* dy, int64_t M, int64_t N, int64_t K, float* dx, float* dw, float* db,
                    stlt_ctx* ctx, void* scratch, size_t scratch_bytes, stlt_stream_t stream);
/* Backward of K3 on the packed projection (what the training sweep runs per layer): dqkv (S*L, 3*H*dh) from qkv and dctx, with
 * the masks of stlt_attn_core_fwd, optional dropout of the probabilities (site as in stlt_train_forward) and, when in_proj_b_grad
 * is not NULL, in_proj_b_grad (3*H*dh) += column sums of dqkv.  dh == 64, L <= 64: v_mfma_f32_16x16x4_f32 tiles (one wave per
 * sequence and head); up to 256: plain FMA, keys streamed through LDS.  Any other head dim (1 ... 256): the two-phase vector-ALU
 * kernel of csrc/attn_any.hip (L <= 256, no atomics).  scratch: stlt_attn_core_bwd_scratch_bytes(H) bytes. */
size_t stlt_attn_core_bwd_scratch_bytes(int64_t H);
int stlt_attn_core_bwd(const float* qkv, const float* dctx, const uint8_t* kpm, int causal, int64_t S, int64_t L, int64_t H, int64_t dh,
                       float dropout_p, uint64_t seed, uint32_t site, float* dqkv, float* in_proj_b_grad, void* scratch, size_t scratch_bytes,
                       stlt_stream_t stream);
/* attention with train-mode dropout of the probabilities (nn.MultiheadAttention(dropout=p)): counter-based mask from
 * (seed, site, query row, head, key position); stlt_attn_bwd given the same (p, seed, site) recomputes it.  p = 0: none. */
int stlt_attn_fwd_dropout(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kpm, int causal,
                          int64_t S, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float dropout_p, uint64_t seed, uint32_t site, float* ctx,
                          stlt_stream_t stream);
int stlt_attn_bwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* dctx, const uint8_t* kpm,
                  int causal, int64_t S, int64_t Lq, int64_t Lk, int64_t H, int64_t dh, float dropout_p, uint64_t seed, uint32_t site,
                  float* dq, int64_t lddq, float* dk, float* dv, int64_t lddkv, stlt_stream_t stream);
size_t stlt_add_layernorm_bwd_scratch_bytes(int64_t d);
int stlt_add_layernorm_bwd(const float* dy, const float* x, const float* res, const float* ln_w, float eps, int64_t M, int64_t d,
                           float* ds, float* g_w, float* g_b, void* scratch, size_t scratch_bytes, stlt_stream_t stream);
int stlt_gelu_fwd(const float* u, float* h, int64_t n, stlt_stream_t stream);
/* K1 / K7 for an op-level autograd: the forward also returns the pre-LayerNorm sum; the LayerNorm part of the backward is
 * stlt_add_layernorm_bwd on that sum (res = NULL), the rest is below: given d_pre (gradient wrt the sum) the parameter
 * gradients ACCUMULATE into g_* (nullable); the gradient wrt the K7 input rows is d_pre itself.  Row 0 of the category /
 * frame-type tables is the padding index and receives none (models.py:22,91). */
int stlt_embed_fwd_train(const int64_t* categories, const float* boxes, const float* scores, const float* cat_table, int64_t n_categories,
                         const float* box_w, const float* box_b, const float* score_w, const float* score_b, const float* ln_w,
                         const float* ln_b, float eps, int64_t n_tokens, int64_t d, float* pre_out, float* out, stlt_stream_t stream);
size_t stlt_embed_bwd_scratch_bytes(int64_t n_tokens, int64_t n_categories, int64_t d);
int stlt_embed_bwd(const float* d_pre, const int64_t* categories, const float* boxes, const float* scores, int64_t n_categories,
                   int64_t n_tokens, int64_t d, float* g_cat, float* g_box_w, float* g_box_b, float* g_score_w, float* g_score_b,
                   void* scratch, size_t scratch_bytes, stlt_stream_t stream);
int stlt_frames_embed_fwd_train(const float* spatial, int64_t row_stride, const int64_t* frame_types, const float* pos_table,
                                const float* type_table, const float* ln_w, const float* ln_b, float eps, int64_t B, int64_t T, int64_t d,
                                float* pre_out, float* out, stlt_stream_t stream);
size_t stlt_frames_embed_bwd_scratch_bytes(int64_t T, int64_t d);
int stlt_frames_embed_bwd(const float* d_pre, const int64_t* frame_types, int64_t B, int64_t T, int64_t d, float* g_pos, float* g_type,
                          void* scratch, size_t scratch_bytes, stlt_stream_t stream);
int stlt_gelu_bwd(const float* dh, const float* u, float* du, int64_t n, stlt_stream_t stream);
/* Element-wise pieces of the op-level training path.  stlt_dropout: y[i] = keep(seed, site, i) ? x[i] / (1-p) : 0 with the
 * library's counter-based mask — the backward is the same call on the gradient (nn.Dropout at models.py:333-376 and inside
 * nn.TransformerEncoderLayer).  stlt_relu_bwd: dx = dy where the activation's OUTPUT y is positive (the ReLU itself runs in
 * the producing product's epilogue: stlt_linear_fwd with STLT_ACT_RELU).  Buffers 16-byte aligned; y may alias x / dx may alias dy. */
int stlt_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint32_t site, stlt_stream_t stream);
int stlt_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, stlt_stream_t stream);

/* Criterion of the reference (utils/train_inference_utils.py:64-76) and its gradient in one pass: loss_out[0] = weight *
 * mean loss, dlogits = weight * d(mean loss)/d(logits).  CROSS_ENTROPY: labels int64 (B); BCE_WITH_LOGITS: labels float
 * (B,K) multi-hot.  `weight` = 1 / number of logit heads (the reference averages the heads' losses).  scratch: >= B floats. */
#define STLT_LOSS_CROSS_ENTROPY 0
#define STLT_LOSS_BCE_WITH_LOGITS 1
int stlt_loss_fwd_bwd(const float* logits, const void* labels, int kind, int64_t B, int64_t K, float weight,
                      float* scratch, float* loss_out, float* dlogits, stlt_stream_t stream);
int stlt_grad_norm(const float* flat_grad, int64_t n, float max_norm, float* scratch, float* out, stlt_stream_t stream);
int stlt_adamw_step(const stlt_opt_chunk* chunks_dev, int64_t n_chunks, const float* flat_grad, float* exp_avg, float* exp_avg_sq,
                    const float* norm_and_clip, float lr, float beta1, float beta2, float eps, int64_t step, stlt_stream_t stream);

/* ---- block-level training calls for the fusion models (CAF / CACNF / LCF; reference models.py:328-431, 239-246) ----
 * One native forward and one native reverse call per residual block, so an optimisation step of a fusion model is a few
 * dozen calls instead of a few hundred op-level ones.  The caller owns the tape tensors (q / kv / ctx / a, u / h / f) and the
 * two backward buffers (stlt_block_keep_bytes / stlt_block_work_bytes, below).
 *   attention block (SelfAttentionLayer / CrossAttentionLayer :345-382, and the first half of nn.TransformerEncoderLayer):
 *     out = LN_eps(x + drop(MHA(x, c, c) Wo^T + bo)); c == NULL: self-attention on a packed q|k|v buffer q (S*Lq, 3d);
 *     else q (S*Lq, d) and kv (S*Lk, 2d).  kpm (S*Lk) bytes over the keys (never NULL: pass zeros).  Dropout sites: site0 =
 *     attention probabilities, site0 + 1 = in front of the residual.
 *   feed-forward block (:384-401 layout_ffn, and the second half of nn.TransformerEncoderLayer):
 *     out = LN_eps(x + drop(W2 drop_inner(act(W1 x + b1)) + b2)), act = STLT_ACT_GELU (u and h kept) or STLT_ACT_RELU (h kept);
 *     inner_dropout != 0 applies site0 to the hidden, site0 + 1 is the dropout in front of the residual.
 * gemm_scratch (forward calls): NULL, or stlt_gemm_scratch_bytes() of device memory lent for the call's stream-K launches.
 * The backward calls ACCUMULATE (+=) into the gradient struct's buffers (NULL members are skipped), write dx (and dc, the
 * gradient wrt the context tokens of a cross-attention block), and recompute the dropout masks from (drop_p, seed, site0). */
/* Deferred weight gradients of the block calls.  After stlt_ctx_dw_defer(ctx, 1) the *_block_bwd_train calls that name `ctx` queue their
 * weight-gradient products (g_w += dyᵀ·x) in the context instead of launching them; stlt_ctx_dw_flush runs the queue as grouped stream-K
 * launches of up to 32 products on `stream` (the stream the blocks ran on) with stlt_gemm_scratch_bytes() of scratch, in queue order (two
 * products into the same gradient never share a launch).  The CALLER keeps every operand of the queued products alive and unchanged until the
 * flush: the blocks' `keep` buffers (they hold the output gradients) and the forward activations the blocks were handed — their `work`
 * buffers are free again when the call returns.  mode 0 stops collecting (queued products stay), -1 stops and discards.  The queue holds at
 * most 512 products: a block that would overflow it launches its own.  A training step of the fusion models (models.py:403-431) makes 34
 * block calls.  torch's autograd engine runs the block backwards on its own thread: the queue follows the handle, not the thread. */
int stlt_ctx_dw_defer(stlt_ctx* ctx, int mode);
int stlt_ctx_dw_pending(stlt_ctx* ctx);
int stlt_ctx_dw_flush(stlt_ctx* ctx, void* gemm_scratch, size_t gemm_scratch_bytes, stlt_stream_t stream);
/* Transposed weight copies for the input-gradient products of a training step (csrc/wt_cache.hip).  dX = dY·W with W (n_out, k_in) read as
 * it lies runs 13 - 17 % below a forward product of the same shape on the small-tile kernel; with a copy wt (k_in, n_out) it IS a forward
 * product.  stlt_ctx_wt_refresh writes every entry's copy (wt[k][n] = w[n][k]; caller-owned buffers; dimensions multiples of 4, 16-byte
 * aligned) on `stream` and makes the set current in `ctx`: until stlt_ctx_wt_clear, every input-gradient product of a call that names
 * `ctx` on the same device (stlt_train_backward, the block backwards, stlt_linear_bwd, stlt_input_grad_small with tile 0) whose weight
 * pointer lies inside a registered weight — row ranges of a packed in-projection starting at a multiple of four rows included — and that
 * routes to the small tiles reads the copy.  The library orders each such call's stream behind the transposes.  The caller refreshes after
 * the weights changed (train.Trainer: at the start of every step) and clears before anything else may change them (at the end of the
 * step); a call that names another context, or none, never sees a copy.  stlt_ctx_wt_hits: products LAUNCHED on a copy so far. */
typedef struct {
  const float* w;      /* (n_out, k_in) row-major: nn.Linear's weight */
  float* wt;           /* (k_in, n_out) row-major: the copy */
  int64_t n_out, k_in;
} stlt_wt_entry;
int stlt_ctx_wt_refresh(stlt_ctx* ctx, const stlt_wt_entry* entries, int64_t n, stlt_stream_t stream);
int stlt_ctx_wt_clear(stlt_ctx* ctx);
long long stlt_ctx_wt_hits(stlt_ctx* ctx);
/* Device memory of a block backward, in two buffers: `keep` (stlt_block_keep_bytes(rows, d, kind); kind 0 = attention block, 1 =
 * feed-forward block; rows = the larger token count of the block) holds the gradients the block's weight-gradient products read — with
 * deferral on it must stay untouched until the flush, one buffer per block call; `work` (stlt_block_work_bytes(rows, d): stream-K partial
 * tiles, reduction pools, a context gradient) is dead when the call has enqueued its launches — one buffer per stream serves all blocks. */
size_t stlt_block_keep_bytes(int64_t rows, int64_t d, int kind);
size_t stlt_block_work_bytes(int64_t rows, int64_t d);
int stlt_attn_block_fwd_train(const stlt_attn_block_params* p, int64_t d, int64_t H, float eps, const float* x, int64_t Lq, const float* c,
                              int64_t Lk, const uint8_t* kpm, int causal, int64_t S, float drop_p, uint64_t seed, uint32_t site0, float* q,
                              float* kv, float* ctx, float* a, float* out, void* gemm_scratch, size_t gemm_scratch_bytes, stlt_stream_t stream);
int stlt_attn_block_bwd_train(const stlt_attn_block_params* p, const stlt_attn_block_params* g, int64_t d, int64_t H, float eps, const float* x,
                              int64_t Lq, const float* c, int64_t Lk, const uint8_t* kpm, int causal, int64_t S, float drop_p, uint64_t seed,
                              uint32_t site0, const float* q, const float* kv, const float* ctx, const float* a, const float* dy, float* dx,
                              float* dc, stlt_ctx* tctx, void* keep, size_t keep_bytes, void* work, size_t work_bytes, stlt_stream_t stream);
int stlt_ffn_block_fwd_train(const stlt_ffn_block_params* p, int64_t d, float eps, int act, int inner_dropout, const float* x, int64_t M,
                             float drop_p, uint64_t seed, uint32_t site0, float* u, float* h, float* f, float* out, void* gemm_scratch,
                             size_t gemm_scratch_bytes, stlt_stream_t stream);
int stlt_ffn_block_bwd_train(const stlt_ffn_block_params* p, const stlt_ffn_block_params* g, int64_t d, float eps, int act, int inner_dropout,
                             const float* x, int64_t M, float drop_p, uint64_t seed, uint32_t site0, const float* u, const float* h, const float* f,
                             const float* dy, float* dx, stlt_ctx* tctx, void* keep, size_t keep_bytes, void* work, size_t work_bytes, stlt_stream_t stream);

/* ---- evaluators on the device (reference src/utils/evaluation.py) ----
 * stlt_eval_topk: EvaluatorSomething.process (evaluation.py:21-34) for one logit head.  counts[0] += clips whose label is
 * the arg-max, counts[1] += clips whose label is among the five largest logits (int64 device counters, accumulated with
 * integer atomics).  The label's rank is the number of classes with a larger logit, or an equal logit at a lower index.
 * stlt_eval_average_precision: map() + the empty-clip rule of charades_map (evaluation.py:100-131).  scores / truths are
 * (n, C) row-major float (scores = sigmoid of the logits, truths multi-hot); ap[c] = average precision of class c (NaN
 * when the class has no positive), positives[c] = its number of positives; scratch >= n bytes.  n <= stlt_eval_max_clips()
 * (a class column is sorted inside one workgroup's LDS).  Equal scores keep clip order (the reference's argsort leaves
 * ties unspecified). */
int stlt_eval_topk(const float* logits, int64_t ld, const int64_t* labels, int64_t B, int64_t K, int64_t* counts,
                   stlt_stream_t stream);
int64_t stlt_eval_max_clips(void);
/* EvaluatorActionGenome.process (evaluation.py:76-82) for one batch: pred[i][c] = (double)sigmoid_f32(logits[i][c]) and
 * truth[i][c] = (double)labels[i][c] written into rows [row0, row0 + B) of the two (total, C) float64 device tables. */
int stlt_eval_store_sigmoid(const float* logits, int64_t ld, const float* labels, int64_t B, int64_t C, double* pred, double* truth,
                            int64_t row0, stlt_stream_t stream);
int stlt_eval_average_precision(const float* scores, const float* truths, int64_t n, int64_t C, double* ap, double* positives,
                                uint8_t* scratch, stlt_stream_t stream);

/* ---- per-kernel timing (bench.py roofline leg): hipEvents around every launch of the whole-path calls ---- */
#define STLT_K_EMBED 0
#define STLT_K_GEMM 1
#define STLT_K_ATTN_SPATIAL 2
#define STLT_K_ATTN_TEMPORAL 3
#define STLT_K_ADDLN 4
#define STLT_K_FRAMES 5
#define STLT_K_GATHER 6
#define STLT_K_LN_BWD 7        /* LayerNorm backward (+ its partial-row reductions) */
#define STLT_K_ATTN_BWD 8      /* attention backward */
#define STLT_K_GELU 9          /* stand-alone GELU forward (training tape) / backward */
#define STLT_K_EMBED_BWD 10    /* embedding / frames-embedding backward */
#define STLT_K_OPTIM 11        /* criterion, gradient norm, AdamW */
#define STLT_K_MISC 12         /* row gathers / scatters, column sums, ragged index, the head's small products */
#define STLT_K_MHSA_FUSED 13    /* fused in-projection + causal attention core (stlt_mhsa_fused_fwd): temporal tower */
#define STLT_K_MHSA_FUSED_SPATIAL 14  /* the same kernel's non-causal launches (spatial tower) */
#define STLT_K_COUNT 15
/* The switch is process-wide; the records (and both calls below) belong to the device that is CURRENT when they are made: a
 * process driving several GPUs collects once per device (with that device current) before it switches timing off, or the
 * other devices' records stay queued. */
int stlt_prof_enable(int on);                       /* 1: record events around each launch (serialises nothing, adds events) */
int stlt_prof_collect(double* ms_out, int64_t* launches_out);  /* sync events, accumulate per-kernel ms / launch counts (STLT_K_COUNT entries each), reset */
/* FLOPs (2*M*N*K summed over the launches) of the matrix-core products enqueued on the current device while timing was on,
 * since the last call: what the roofline of a step is priced with, whatever the schedule (forward, elided layers, backward). */
double stlt_prof_take_gemm_flops(void);
/* The same records launch by launch, in launch order (alternative to stlt_prof_collect: either call drains them): duration from the
 * launch's two events, the FLOPs / algorithmic bytes the launcher declared, and the launcher's note — shape, tile, workgroups, rounds,
 * k-steps — for tools/launch_bound.py, which prices every launch of a step against its own bound.  *n_out = records drained (may exceed
 * cap: the first cap are written). */
#define STLT_PROF_NOTE 160
typedef struct { int kid; int kernels; float us; float reserved; double flops, bytes; char note[STLT_PROF_NOTE]; } stlt_prof_launch;  /* kernels: kernel launches inside the record (a stream-K product = 2) */
int stlt_prof_launches(stlt_prof_launch* out, int64_t cap, int64_t* n_out);

/* Diagnostics only (tools/attn_stamps.py, tools/gemm_block_times.py): when non-NULL, stlt_attn_core_fwd runs its
 * s_memtime-stamped build (8 uint64 phase stamps per item) and stlt_linear_fwd records per-workgroup data into dev_buf:
 * 4 uint64 per workgroup (start / end on the 100 MHz clock, XCC id, tile count), then 48 uint64 per workgroup of per-wave
 * phase stamps (STLT_GEMM_STAMP builds), 1024 spare words and 2 uint64 per workgroup of shader-clock start / end — the
 * GEMM writes the first and the last region whenever the buffer is set, so it must hold stlt_debug_buffer_bytes().
 * Pass NULL to restore normal operation. */
int stlt_debug_set_buffer(void* dev_buf);
size_t stlt_debug_buffer_bytes(void);  /* (54 * compute units of the current device + 1024) * 8 */

/* Opt-in "split-bf16" products (csrc/gemm_bf16x3.hip): every f32 operand element is cut into three bf16 pieces and a product is
 * the six piece products of weight >= 2^-24 on v_mfma_f32_16x16x32_bf16, accumulated in f32 — the dropped terms are below the
 * rounding of an f32 multiply.  terms = 6 turns it on, 0 off; the STLT_GEMM_SPLIT_BF16=6 environment variable is the initial
 * value.  What then runs on it: every nn.Linear forward of the entry points above (stlt_linear_fwd, the whole-path forwards,
 * the training forward, the block calls) and the input-gradient products dX = dY·W of stlt_train_backward and the block
 * backwards (through a transposed copy of W in their scratch), when the launch has whole tiles filling at least half of the
 * workgroups (STLT_X3_MIN_FILL, default 0.5), K % 32 == 0 and K >= 64; everything else — the weight-gradient products, small
 * launches — keeps the f32-MFMA kernel.  Not the default: results agree with the f32 kernel's to f32 rounding (error against an
 * fp64 product within eps_f32 * sqrt(K) of the largest output, the bound of one sequential f32 accumulation: equal to the f32
 * kernel's within 15 % on whole-tile launches, up to ~9x it on small ones, whose stream-K form sums K in short ranges) but are not
 * bit-identical to it; an output is finite exactly where the f32 kernel's is, but infinite operands give NaN where the f32 kernel
 * gives +-inf; and bench.py never reports it as `value` (side objects of the JSON line).  Process-wide switch;
 * STLT_EINVAL for other values. */
int stlt_set_gemm_split_bf16(int terms);

#ifdef __cplusplus
}
#endif
#endif /* STLT_HIP_H */
