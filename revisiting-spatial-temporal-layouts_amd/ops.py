"""Tensor-level wrappers over the C-ABI (one function per entry point of include/stlt_hip.h).

PyTorch is plumbing here: device allocations, the current HIP stream, nothing else.  Every wrapper checks
that its tensors live on a GPU, are fp32/int64/bool as the ABI expects and contiguous, then passes raw
device pointers.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional

import torch

from . import _lib as L


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise L.StltHipError(f"{name}: expected a GPU tensor (the STLT hot path has no CPU fallback), got "
                             f"{getattr(t, 'device', type(t))}")
    if t.dtype != dtype:
        raise L.StltHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise L.StltHipError(f"{name}: tensor must be contiguous")
    return t


def _mask_u8(t: torch.Tensor, name: str) -> torch.Tensor:
    """bool masks are 1 byte/element: reinterpret, never copy."""
    if t.dtype == torch.bool:
        t = t.contiguous().view(torch.uint8)
    return _chk(t, torch.uint8, name)


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def embed(categories, boxes, scores, cat_table, box_w, box_b, score_w, score_b, ln_w, ln_b, eps: float):
    """K1 — CategoryBoxEmbeddings.forward (reference src/modelling/models.py:29-39). -> (*categories.shape, d)"""
    lib = L.load()
    _chk(categories, torch.int64, "categories"); _chk(boxes, torch.float32, "boxes")
    if scores is not None:
        _chk(scores, torch.float32, "scores")
    for n, t in (("cat_table", cat_table), ("box_w", box_w), ("box_b", box_b), ("ln_w", ln_w), ("ln_b", ln_b)):
        _chk(t, torch.float32, n)
    d = cat_table.shape[1]
    n_tok = categories.numel()
    assert boxes.numel() == n_tok * 4
    out = torch.empty(*categories.shape, d, device=categories.device, dtype=torch.float32)
    L.check(lib.stlt_embed_fwd(_p(categories), _p(boxes), _p(scores), _p(cat_table), cat_table.shape[0], _p(box_w),
                               _p(box_b), _p(score_w), _p(score_b), _p(ln_w), _p(ln_b), eps, n_tok, d, _p(out),
                               _stream()), "stlt_embed_fwd")
    return out


_TLS = threading.local()
_SKINNY_ROWS = 128  # STLT_GEMM_SKINNY_ROWS' default: products of at most this many rows can run as split-k partial tiles when scratch is lent


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int = L.ACT_NONE,
           out: Optional[torch.Tensor] = None, rows: Optional[int] = None, ldx: Optional[int] = None):
    """nn.Linear forward y = act(x wᵀ + b) on the f32 matrix cores.  x (..., K) contiguous; with rows/ldx a strided row view."""
    lib = L.load()
    _chk(x, torch.float32, "x"); _chk(w, torch.float32, "w")
    if bias is not None:
        _chk(bias, torch.float32, "bias")
    N, K = w.shape
    if rows is None:
        assert x.shape[-1] == K
        M, ldx_ = x.numel() // K, K
        out_shape = (*x.shape[:-1], N)
    else:
        M, ldx_ = rows, ldx
        out_shape = (rows, N)
    if out is None:
        out = torch.empty(out_shape, device=x.device, dtype=torch.float32)
    lend = 0 < M <= _SKINNY_ROWS and x.is_cuda and not getattr(_TLS, "user_scratch", False)
    if lend:
        # a few rows (the heads of a 64-clip step): with scratch lent the library runs them as split-k partial tiles over the whole chip
        # instead of 2 - 12 workgroups of serial k-steps (include/stlt_hip.h: stlt_gemm_set_scratch; csrc/gemm_any.hip: launch_gemm_skinny)
        sk = _sk_scratch(x.device)
        L.check(lib.stlt_gemm_set_scratch(sk.data_ptr(), sk.numel()), "stlt_gemm_set_scratch")
    try:
        L.check(lib.stlt_linear_fwd(_p(x), ldx_, _p(w), _p(bias), _p(out), N, M, N, K, act, _stream()), "stlt_linear_fwd")
    finally:
        if lend:
            lib.stlt_gemm_set_scratch(None, 0)
    return out


def small_tile(tile_cols: int, tile_rows: int = 128) -> int:
    """The C-ABI's tile parameter: columns | rows << 16, 128 rows as plain columns (include/stlt_hip.h: stlt_linear_small_fwd)."""
    if int(tile_cols) >> 16:  # already in the C-ABI's form (what stlt_linear_small_choice returns)
        return int(tile_cols)
    return int(tile_cols) if int(tile_rows) == 128 else (int(tile_cols) | (int(tile_rows) << 16))


SMALL_TILES = tuple((128, c) for c in (48, 64, 96, 128, 144, 192)) + tuple((64, c) for c in (64, 96, 128, 160, 192, 256)) + tuple((32, c) for c in (128, 192, 256))


def linear_small(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], tile_cols: int, act: int = L.ACT_NONE,
                 residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, tile_rows: int = 128):
    """The same product on the small-tile kernel (csrc/gemm16.hip; include/stlt_hip.h: stlt_linear_small_fwd): whole tiles of tile_rows x
    tile_cols (one of SMALL_TILES); `residual` (M, N) is added after the bias (act must then be ACT_NONE)."""
    lib = L.load()
    _chk(x, torch.float32, "x"); _chk(w, torch.float32, "w")
    N, K = w.shape
    M = x.numel() // K
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    L.check(lib.stlt_linear_small_fwd(_p(x), K, _p(w), _p(bias), _p(residual), N, _p(out), N, M, N, K, act, small_tile(tile_cols, tile_rows), _stream()),
            "stlt_linear_small_fwd")
    return out


def input_grad_small(dy: torch.Tensor, w: torch.Tensor, tile_cols: int, residual: Optional[torch.Tensor] = None, tile_rows: int = 128,
                     context: Optional["TrainContext"] = None):
    """dx (M, k_in) = dy (M, n_out) · w (n_out, k_in) (+ residual) on the small-tile kernel (include/stlt_hip.h: stlt_input_grad_small):
    on the tile given, the weight read as it lies; tile_cols = 0 routes by the launch-time estimate and reads `context`'s transposed copy
    of w when it holds a current one."""
    lib = L.load()
    _chk(dy, torch.float32, "dy"); _chk(w, torch.float32, "w")
    n_out, k_in = w.shape
    M = dy.numel() // n_out
    dx = torch.empty(*dy.shape[:-1], k_in, device=dy.device, dtype=torch.float32)
    tile = 0 if tile_cols == 0 else small_tile(tile_cols, tile_rows)
    L.check(lib.stlt_input_grad_small(_p(dy), n_out, _p(w), n_out, k_in, _p(residual), k_in, _p(dx), k_in, M, tile, _ctx_handle(context), _stream()),
            "stlt_input_grad_small")
    return dx


def set_train_side_stream(on) -> None:
    """stlt_train_backward's weight-gradient products on the library's side stream (default) or on the caller's stream; None: back to
    STLT_TRAIN_DW_STREAM / the default."""
    L.check(L.load().stlt_set_train_side_stream(-1 if on is None else int(bool(on))), "stlt_set_train_side_stream")


def get_train_side_stream() -> bool:
    """The setting in force (what an A/B restores afterwards)."""
    return bool(L.load().stlt_get_train_side_stream())


def set_gemm_small_tiles(mode: int) -> None:
    """Routing of under-filled products to the small-tile kernel: -1 by estimate (default), 0 off, 1 always, 128 / 64 / 32 by estimate over tiles
    of that height only, -2 back to the process's initial
    setting (STLT_GEMM16 or the default) — what to restore after an A/B (stlt_set_gemm_small_tiles)."""
    L.check(L.load().stlt_set_gemm_small_tiles(int(mode)), "stlt_set_gemm_small_tiles")


def set_gemm_split_bf16(terms: int) -> None:
    """Opt-in split-bf16 build of the `linear` forward (include/stlt_hip.h: stlt_set_gemm_split_bf16): 6 = six bf16 MFMA
    products per f32 product, f32-equivalent results; 0 = the f32-MFMA kernel (default).  Process-wide."""
    L.check(L.load().stlt_set_gemm_split_bf16(int(terms)), "stlt_set_gemm_split_bf16")


class gemm_scratch:
    """Context manager: lend the calling thread's `linear` / `gemm` launches a scratch buffer so that under-filled
    launches run as stream-K (include/stlt_hip.h: stlt_gemm_set_scratch).  The whole-path calls do this themselves."""

    def __init__(self, device="cuda"):
        self.device = device
        self.buf = None

    def __enter__(self):
        lib = L.load()
        n = lib.stlt_gemm_scratch_bytes()
        self.buf = torch.empty(n, dtype=torch.uint8, device=self.device)
        L.check(lib.stlt_gemm_set_scratch(self.buf.data_ptr(), n), "stlt_gemm_set_scratch")
        _TLS.user_scratch = True  # `linear` leaves the calling thread's lent scratch alone
        return self

    def __exit__(self, *exc):
        _TLS.user_scratch = False
        L.check(L.load().stlt_gemm_set_scratch(None, 0), "stlt_gemm_set_scratch")
        torch.cuda.synchronize()  # the buffer may still be read by enqueued fix-up kernels
        self.buf = None
        return False


def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False,
         add: Optional[torch.Tensor] = None, n_split: int = 1, k: Optional[int] = None):
    """c = opA(a) @ opB(b) (+ add) on the f32-MFMA kernel (backward layouts of nn.Linear).  a: (M,K) or (K,M) if trans_a;
    b: (N,K) or (K,N) if trans_b.  With n_split > 1 the contraction is split and summed by the slab reduction.
    `k` overrides the contraction length (rows of contraction-major operands beyond the logical K must be zero)."""
    lib = L.load()
    _chk(a, torch.float32, "a"); _chk(b, torch.float32, "b")
    M = a.shape[1] if trans_a else a.shape[0]
    N = b.shape[1] if trans_b else b.shape[0]
    K = k if k is not None else (a.shape[0] if trans_a else a.shape[1])
    lda, ldb = a.shape[1], b.shape[1]
    if add is not None:
        _chk(add, torch.float32, "add")
    if n_split == 1:
        c = torch.empty(M, N, device=a.device, dtype=torch.float32)
        L.check(lib.stlt_gemm(int(trans_a), int(trans_b), _p(a), lda, _p(b), ldb, _p(add), N, _p(c), N, 0, M, N, K, 1,
                              _stream()), "stlt_gemm")
        return c
    slabs = torch.empty(n_split, M, N, device=a.device, dtype=torch.float32)
    L.check(lib.stlt_gemm(int(trans_a), int(trans_b), _p(a), lda, _p(b), ldb, None, 0, _p(slabs), N, M * N, M, N, K,
                          n_split, _stream()), "stlt_gemm")
    c = add.clone() if add is not None else torch.empty(M, N, device=a.device, dtype=torch.float32)
    L.check(lib.stlt_reduce_slabs(_p(slabs), M * N, n_split, _p(c), M * N, int(add is not None), _stream()),
            "stlt_reduce_slabs")
    return c


def weight_grad_group(items):
    """g_w += dy[:rows]ᵀ @ x[:rows] for every (dy, x, g_w[, rows]) of `items`, as one grouped stream-K launch (needs
    `gemm_scratch()`); rows defaults to dy.shape[0] and must be a multiple of 32."""
    lib = L.load()
    arr = (L.WgradItem * len(items))()
    for i, it in enumerate(items):
        dy, x, g = it[0], it[1], it[2]
        rows = it[3] if len(it) > 3 else dy.shape[0]
        _chk(dy, torch.float32, "dy"); _chk(x, torch.float32, "x"); _chk(g, torch.float32, "g_w")
        assert tuple(g.shape) == (dy.shape[1], x.shape[1]) and x.shape[0] >= rows and dy.shape[0] >= rows
        arr[i] = L.WgradItem(_p(dy), dy.shape[1], _p(x), x.shape[1], rows, _p(g))
    L.check(lib.stlt_weight_grad_group(arr, len(items), _stream()), "stlt_weight_grad_group")


def attn_core_bwd(qkv: torch.Tensor, dctx: torch.Tensor, kpm: torch.Tensor, causal: bool, num_heads: int, dropout_p: float = 0.0, seed: int = 0,
                  site: int = 0, want_bias_grad: bool = False):
    """Backward of K3 on the packed projection (include/stlt_hip.h: stlt_attn_core_bwd): qkv (S,L,3d), dctx (S,L,d) -> dqkv (S,L,3d)
    [, column sums of dqkv (3d)]."""
    lib = L.load()
    _chk(qkv, torch.float32, "qkv"); _chk(dctx, torch.float32, "dctx")
    kpm = _mask_u8(kpm, "kpm")
    S, Lq, d3 = qkv.shape
    dqkv = torch.empty_like(qkv)
    gb = torch.zeros(d3, device=qkv.device, dtype=torch.float32) if want_bias_grad else None
    sc = _scratch(int(lib.stlt_attn_core_bwd_scratch_bytes(num_heads)), qkv.device)
    L.check(lib.stlt_attn_core_bwd(_p(qkv), _p(dctx), _p(kpm), int(bool(causal)), S, Lq, num_heads, d3 // 3 // num_heads, float(dropout_p), int(seed),
                                   int(site), _p(dqkv), _p(gb), sc.data_ptr(), sc.numel(), _stream()), "stlt_attn_core_bwd")
    return (dqkv, gb) if want_bias_grad else dqkv


def mhsa_fused(x: torch.Tensor, in_proj_w: torch.Tensor, in_proj_b: torch.Tensor, kpm: torch.Tensor, num_heads: int, causal: bool = True,
               want_qkv: bool = False, dropout_p: float = 0.0, seed: int = 0, site: int = 0):
    """Fused in-projection + attention core: x (S,L,d), kpm (S,L) -> ctx (S,L,d); L <= 64.  want_qkv (or dropout_p > 0): the training
    form, returns (ctx, qkv) with the packed projections (S,L,3d) the reverse sweep needs."""
    lib = L.load()
    _chk(x, torch.float32, "x"); _chk(in_proj_w, torch.float32, "in_proj_w"); _chk(in_proj_b, torch.float32, "in_proj_b")
    kpm = _mask_u8(kpm, "kpm")
    S, Lq, d = x.shape
    ctx = torch.empty_like(x)
    if not (want_qkv or dropout_p > 0.0) and causal:
        L.check(lib.stlt_mhsa_fused_fwd(_p(x), _p(in_proj_w), _p(in_proj_b), _p(kpm), S, Lq, num_heads, d, _p(ctx), _stream()), "stlt_mhsa_fused_fwd")
        return ctx
    qkv = torch.empty(S, Lq, 3 * d, device=x.device, dtype=torch.float32) if (want_qkv or dropout_p > 0.0) else None
    L.check(lib.stlt_mhsa_fused_fwd_ex(_p(x), _p(in_proj_w), _p(in_proj_b), _p(kpm), int(bool(causal)), S, Lq, num_heads, d, float(dropout_p), int(seed),
                                       int(site), _p(ctx), _p(qkv), _stream()), "stlt_mhsa_fused_fwd_ex")
    return (ctx, qkv) if qkv is not None else ctx


def attn_core(qkv: torch.Tensor, kpm: torch.Tensor, causal: bool, num_heads: int):
    """K3 — qkv (S,L,3d) packed [q;k;v], kpm (S,L) bool/uint8 (True = key masked). -> ctx (S,L,d)"""
    lib = L.load()
    _chk(qkv, torch.float32, "qkv")
    kpm = _mask_u8(kpm, "kpm")
    S, Lq, d3 = qkv.shape
    d = d3 // 3
    ctx = torch.empty(S, Lq, d, device=qkv.device, dtype=torch.float32)
    L.check(lib.stlt_attn_core_fwd(_p(qkv), _p(kpm), int(bool(causal)), S, Lq, num_heads, d // num_heads, _p(ctx),
                                   _stream()), "stlt_attn_core_fwd")
    return ctx


def attn_cross(q: torch.Tensor, kv: torch.Tensor, kpm_k: Optional[torch.Tensor], num_heads: int, causal: bool = False):
    """Cross-attention core: q (S,Lq,d) projected queries, kv (S,Lk,2d) packed [k;v], kpm_k (S,Lk) over the keys or None."""
    lib = L.load()
    _chk(q, torch.float32, "q"); _chk(kv, torch.float32, "kv")
    S, Lq, d = q.shape
    Lk = kv.shape[1]
    if kpm_k is None:
        kpm_k = torch.zeros(S, Lk, dtype=torch.uint8, device=q.device)
    kpm_k = _mask_u8(kpm_k, "kpm")
    ctx = torch.empty(S, Lq, d, device=q.device, dtype=torch.float32)
    L.check(lib.stlt_attn_cross_fwd(_p(q), d, _p(kv), kv.data_ptr() + 4 * d, 2 * d, _p(kpm_k), int(bool(causal)), S, Lq, Lk,
                                    num_heads, d // num_heads, _p(ctx), _stream()), "stlt_attn_cross_fwd")
    return ctx


def attn_ragged(qkv: torch.Tensor, seg_lengths, heads: int, causal: bool = False) -> torch.Tensor:
    """Self-attention over compacted rows cut into consecutive segments of the given lengths (include/stlt_hip.h:
    stlt_attn_ragged_fwd).  qkv: (M, 3*d) packed [q;k;v] rows, M = sum(seg_lengths). -> (M, d)"""
    lib = L.load()
    _chk(qkv, torch.float32, "qkv")
    M, d3 = qkv.shape
    d = d3 // 3
    lens = torch.as_tensor(seg_lengths, dtype=torch.int64)
    assert int(lens.sum()) == M and bool((lens > 0).all())
    ends = torch.cumsum(lens, 0)
    starts = ends - lens
    seg_start = torch.repeat_interleave(starts, lens).to(torch.int32).to(qkv.device)
    seg_end = torch.repeat_interleave(ends, lens).to(torch.int32).to(qkv.device)
    ctx = torch.empty(M, d, device=qkv.device, dtype=torch.float32)
    L.check(lib.stlt_attn_ragged_fwd(_p(qkv), _p(seg_start), _p(seg_end), int(causal), M, heads, d // heads, _p(ctx), _stream()),
            "stlt_attn_ragged_fwd")
    return ctx


def add_layernorm(x: torch.Tensor, res: Optional[torch.Tensor], w: torch.Tensor, b: torch.Tensor, eps: float):
    """out = LayerNorm_eps(x + res) over the last dim (res may be None)."""
    lib = L.load()
    _chk(x, torch.float32, "x"); _chk(w, torch.float32, "ln_w"); _chk(b, torch.float32, "ln_b")
    if res is not None:
        _chk(res, torch.float32, "res")
        assert res.shape == x.shape
    d = x.shape[-1]
    M = x.numel() // d
    out = torch.empty_like(x)
    L.check(lib.stlt_add_layernorm_fwd(_p(x), d, _p(res), d, _p(w), _p(b), eps, M, d, _p(out), d, _stream()),
            "stlt_add_layernorm_fwd")
    return out


def frames_embed(spatial: torch.Tensor, frame_types: torch.Tensor, pos_table, type_table, ln_w, ln_b, eps: float):
    """K7 — spatial (B,T,N,d) (token 0 is read) or (B,T,d); frame_types (B,T). -> (B,T,d)"""
    lib = L.load()
    _chk(spatial, torch.float32, "spatial"); _chk(frame_types, torch.int64, "frame_types")
    B, T = frame_types.shape
    d = spatial.shape[-1]
    row_stride = spatial.numel() // (B * T)
    out = torch.empty(B, T, d, device=spatial.device, dtype=torch.float32)
    L.check(lib.stlt_frames_embed_fwd(_p(spatial), row_stride, _p(frame_types), _p(pos_table), _p(type_table),
                                      _p(ln_w), _p(ln_b), eps, B, T, d, _p(out), _stream()), "stlt_frames_embed_fwd")
    return out


def gather_last(x_btd: torch.Tensor, lengths: torch.Tensor):
    """K8a — out[b] = x[b, lengths[b]-1] (reference models.py:189-192 on the batch-major layout)."""
    lib = L.load()
    _chk(x_btd, torch.float32, "x"); _chk(lengths, torch.int64, "lengths")
    B, T, d = x_btd.shape
    out = torch.empty(B, d, device=x_btd.device, dtype=torch.float32)
    L.check(lib.stlt_gather_last_fwd(_p(x_btd), _p(lengths), B, T, d, _p(out), _stream()), "stlt_gather_last_fwd")
    return out


def workspace_bytes(B: int, T: int, N: int, d: int, n_classes: int) -> int:
    return int(L.load().stlt_workspace_bytes(B, T, N, d, n_classes))


def prof_enable(on: bool):
    L.load().stlt_prof_enable(int(on))


def prof_collect():
    """-> {kernel name: (total ms, launches)} accumulated since the last collect."""
    n = len(L.K_NAMES)
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    L.check(L.load().stlt_prof_collect(ms, cnt), "stlt_prof_collect")
    return {L.K_NAMES[i]: (ms[i], cnt[i]) for i in range(n)}


def prof_launches(cap: int = 65536):
    """-> [{"kernel", "us", "flops", "bytes", "note"}, ...] launch by launch, in launch order, since the last collect (drains the records)."""
    buf = (L.ProfLaunch * cap)()
    n = C.c_int64(0)
    L.check(L.load().stlt_prof_launches(buf, cap, C.byref(n)), "stlt_prof_launches")
    out = []
    for i in range(min(cap, n.value)):
        r = buf[i]
        out.append({"kernel": L.K_NAMES[r.kid] if 0 <= r.kid < len(L.K_NAMES) else str(r.kid), "kernels": int(r.kernels), "us": float(r.us), "flops": float(r.flops),
                    "bytes": float(r.bytes), "note": r.note.decode("utf-8", "replace")})
    return out


def prof_take_gemm_flops() -> float:
    """2*M*N*K summed over the matrix-core launches enqueued while timing was on, since the last call."""
    return float(L.load().stlt_prof_take_gemm_flops())


# ---------------------------------------------------------------------------------------------------------------------
# Op-level autograd over the per-kernel backward entry points (include/stlt_hip.h: stlt_linear_bwd, stlt_attn_bwd,
# stlt_add_layernorm_bwd, stlt_gelu_bwd).  The STLT training step does not use these (it has one fixed reverse sweep);
# they are what the fusion models' training step is composed from (modelling/fusion.py).
_SCRATCH = {}


def _scratch(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    """Grow-only scratch lent to a backward call: one buffer per (device, current stream, slot) — kernels of one stream run in order, so
    consecutive calls may share it; two streams of one device (two training loops on two host threads) must not."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, slot)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        _SCRATCH[key] = buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf


class LinearFn(torch.autograd.Function):
    """y = act(x wᵀ + b), x (..., K); act = ACT_NONE or ACT_RELU (the ReLU of the appearance encoder, applied in the
    product's epilogue; its derivative is taken from the output by ``stlt_relu_bwd``)."""

    @staticmethod
    def forward(ctx, x, w, b, act=L.ACT_NONE):
        if act not in (L.ACT_NONE, L.ACT_RELU):
            raise L.StltHipError("LinearFn: activation must be ACT_NONE or ACT_RELU (GELU has its own Function: the tape keeps the pre-activation)")
        x = x.contiguous()
        y = linear(x, w, b, act=act)
        ctx.save_for_backward(x, w, y if act == L.ACT_RELU else None)
        ctx.bias = b  # the parameter itself (grad_targets looks at its Trainer binding), not saved for its value
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, w, y = ctx.saved_tensors
        N, K = w.shape
        M = x.numel() // K
        dy = dy.contiguous()
        if ctx.act == L.ACT_RELU:
            dz = torch.empty_like(dy)
            L.check(lib.stlt_relu_bwd(_p(dy), _p(y), _p(dz), dy.numel(), _stream()), "stlt_relu_bwd")
            dy = dz
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        # inside a Trainer step the parameter gradients accumulate in place into the bound flat buffer (grad_targets): no zero fill,
        # no AccumulateGrad add (4 launches per Linear; the fusion models' heads and projector are 7 Linears a step)
        (dw, db), grads = grad_targets((w, ctx.bias), (ctx.needs_input_grad[1], ctx.bias is not None and ctx.needs_input_grad[2]))
        nbytes = int(lib.stlt_linear_bwd_scratch_bytes(N))
        sc = _scratch(nbytes, x.device)
        L.check(lib.stlt_linear_bwd(_p(x), _p(w), _p(dy), M, N, K, _p(dx), _p(dw), _p(db), _ctx_handle(context_of((w,))), sc.data_ptr(), sc.numel(),
                                    _stream()), "stlt_linear_bwd")
        return dx, grads[0], grads[1], None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout in train mode with the library's counter-based mask (one seed per call from torch's CPU generator, so
    torch.manual_seed makes a run repeatable); the backward applies the same mask to the gradient."""

    _site = 0x200000  # a site id of its own, apart from AttnFn's; every call draws its own seed, so the id can be constant

    @staticmethod
    def forward(ctx, x, p):
        lib = L.load()
        x = x.contiguous()
        p = float(p)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        site = DropoutFn._site
        y = torch.empty_like(x)
        L.check(lib.stlt_dropout(_p(x), _p(y), x.numel(), p, seed, site, _stream()), "stlt_dropout")
        ctx.meta = (p, seed, site)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        p, seed, site = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.check(lib.stlt_dropout(_p(dy), _p(dx), dy.numel(), p, seed, site, _stream()), "stlt_dropout")
        return dx, None


def dropout(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    """F.dropout's call shape on the native kernel: identity in eval mode or at p = 0."""
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p)


class AttnFn(torch.autograd.Function):
    """softmax(q kᵀ / sqrt(dh) + mask) v per head.  q (S,Lq,d), k / v (S,Lk,d): last dim contiguous, k and v with the same
    row stride (views of a packed projection are fine).  kpm (S,Lk) over the keys or None.  dropout_p > 0: train-mode
    dropout of the attention probabilities with a counter-based mask (seed drawn from torch's CPU generator)."""

    _site = 0x100  # constant: the per-call seed keeps the masks of different attention calls apart, and a forward stays repeatable under torch.manual_seed

    @staticmethod
    def forward(ctx, q, k, v, kpm, causal, heads, dropout_p=0.0):
        lib = L.load()
        S, Lq, d = q.shape
        Lk = k.shape[1]
        for t, name in ((q, "q"), (k, "k"), (v, "v")):
            if t.stride(-1) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
                raise L.StltHipError(f"AttnFn: {name} must be row-strided with a contiguous last dim")
        if k.stride(1) != v.stride(1):
            raise L.StltHipError("AttnFn: k and v need the same row stride")
        kpm8 = torch.zeros(S, Lk, dtype=torch.uint8, device=q.device) if kpm is None else _mask_u8(kpm.contiguous(), "kpm")
        ctxt = torch.empty(S, Lq, d, device=q.device, dtype=torch.float32)
        p = float(dropout_p)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        site = AttnFn._site if p > 0 else 0
        L.check(lib.stlt_attn_fwd_dropout(q.data_ptr(), q.stride(1), k.data_ptr(), v.data_ptr(), k.stride(1), _p(kpm8), int(bool(causal)),
                                          S, Lq, Lk, heads, d // heads, p, seed, site, _p(ctxt), _stream()), "stlt_attn_fwd_dropout")
        ctx.save_for_backward(q, k, v, kpm8)
        ctx.meta = (bool(causal), heads, p, seed, site)
        return ctxt

    @staticmethod
    def backward(ctx, dctx):
        lib = L.load()
        q, k, v, kpm8 = ctx.saved_tensors
        causal, heads, p, seed, site = ctx.meta
        S, Lq, d = q.shape
        Lk = k.shape[1]
        dctx = dctx.contiguous()
        dq = torch.empty(S, Lq, d, device=q.device, dtype=torch.float32)
        dk = torch.empty(S, Lk, d, device=q.device, dtype=torch.float32)
        dv = torch.empty(S, Lk, d, device=q.device, dtype=torch.float32)
        L.check(lib.stlt_attn_bwd(q.data_ptr(), q.stride(1), k.data_ptr(), v.data_ptr(), k.stride(1), _p(dctx), _p(kpm8), int(causal),
                                  S, Lq, Lk, heads, d // heads, p, seed, site, _p(dq), d, _p(dk), _p(dv), d, _stream()), "stlt_attn_bwd")
        return dq, dk, dv, None, None, None, None


class AddLayerNormFn(torch.autograd.Function):
    """LayerNorm_eps(x + res) * w + b over the last dim (res may be None)."""

    @staticmethod
    def forward(ctx, x, res, w, b, eps):
        x = x.contiguous()
        res = None if res is None else res.contiguous()
        ctx.save_for_backward(x, res, w)
        ctx.eps = eps
        ctx.bias = b
        return add_layernorm(x, res, w, b, eps)

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, res, w = ctx.saved_tensors
        d = x.shape[-1]
        M = x.numel() // d
        dy = dy.contiguous()
        ds = torch.empty_like(x)
        (gw, gb), grads = grad_targets((w, ctx.bias), ctx.needs_input_grad[2:4])
        sc = _scratch(int(lib.stlt_add_layernorm_bwd_scratch_bytes(d)), x.device)
        L.check(lib.stlt_add_layernorm_bwd(_p(dy), _p(x), _p(res), _p(w), float(ctx.eps), M, d, _p(ds), _p(gw), _p(gb), sc.data_ptr(),
                                           sc.numel(), _stream()), "stlt_add_layernorm_bwd")
        return ds, (ds if res is not None else None), grads[0], grads[1], None


class GeluFn(torch.autograd.Function):
    """Exact-erf GELU."""

    @staticmethod
    def forward(ctx, u):
        lib = L.load()
        u = u.contiguous()
        if u.numel() % 4:
            raise L.StltHipError("GeluFn: element count must be a multiple of 4")
        h = torch.empty_like(u)
        L.check(lib.stlt_gelu_fwd(_p(u), _p(h), u.numel(), _stream()), "stlt_gelu_fwd")
        ctx.save_for_backward(u)
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = L.load()
        (u,) = ctx.saved_tensors
        dh = dh.contiguous()
        du = torch.empty_like(u)
        L.check(lib.stlt_gelu_bwd(_p(dh), _p(u), _p(du), u.numel(), _stream()), "stlt_gelu_bwd")
        return du


_SK_SCRATCH = {}


def _sk_scratch(device) -> torch.Tensor:
    """Stream-K scratch lent to the block-level forward calls: one buffer per (device, stream) — the grouped stream-K kernels
    write their partial tiles into it and the split-bf16 path its weight planes, so two streams of one device running block
    forwards at the same time must not share it (kept for the life of the process: kernels enqueued on the stream may still be
    reading it)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _SK_SCRATCH.get(key)
    if buf is None:
        _SK_SCRATCH[key] = buf = torch.empty(int(L.load().stlt_gemm_scratch_bytes()), dtype=torch.uint8, device=device)
    return buf



class TrainContext:
    """A training loop's handle into the library (include/stlt_hip.h: stlt_ctx): what the loop leaves inside it between calls — the
    transposed weight copies of the step in progress, the queue of deferred block weight gradients, the side stream + events of its
    reverse sweeps — belongs to this object and to nothing else in the process.  train.Trainer owns one and the backward Functions of the
    models it trains pass it along (context_of); a backward outside a trainer step passes none and gets none of the three."""

    def __init__(self):
        h = C.c_void_p()
        L.check(L.load().stlt_ctx_create(C.byref(h)), "stlt_ctx_create")
        self.handle = h.value
        self.defer_on = False  # inside deferred_block_weight_grads(self)
        self.keep = []         # what the queued products read: the blocks' keep buffers and forward activations, until the flush

    def close(self):
        if getattr(self, "handle", None):
            try:
                L.load().stlt_ctx_destroy(self.handle)
            finally:
                self.handle = None
                self.keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __deepcopy__(self, memo):  # a copied module gets a context of its own, never a second owner of this handle
        return TrainContext()

    def __reduce__(self):
        return (TrainContext, ())

    def wt_hits(self) -> int:
        return int(L.load().stlt_ctx_wt_hits(self.handle))

    def dw_pending(self) -> int:
        return int(L.load().stlt_ctx_dw_pending(self.handle))


def _ctx_handle(context: Optional[TrainContext]):
    return None if context is None else context.handle


def context_of(ws) -> Optional[TrainContext]:
    """The training context a backward call should name: that of the trainer whose step is running its backward over these parameters right
    now (BoundFlatGrads.accumulating), else None — a plain autograd backward never reads a trainer's weight copies or queues into its flush."""
    for w in ws:
        bound = getattr(w, "_stlt_bound", None)
        if bound is not None and bound.accumulating and bound.context is not None:
            return bound.context
    return None


class deferred_block_weight_grads:
    """Context manager around ONE backward pass of a model built from the block Functions (AttnBlockFn / FfnBlockFn), on the current
    stream: the blocks that name `context` queue their weight-gradient products in it instead of launching 2 - 4 of them per block, and the
    exit runs the queue as grouped launches of up to 32 products (include/stlt_hip.h: stlt_ctx_dw_defer / _flush).  Only blocks whose
    parameter gradients are accumulated in place (train.Trainer's bound flat buffer: grad_targets) take part; their keep buffers (the
    gradients the queued products read: 5 - 7 rows x d floats per block, 34 blocks x ~45 MB at 64 clips of CACNF) and forward activations
    stay alive until the flush.  The gradients are complete when the `with` block is left."""

    def __init__(self, context: TrainContext):
        self.context = context

    def __enter__(self):
        lib = L.load()
        L.check(lib.stlt_ctx_dw_defer(self.context.handle, -1), "stlt_ctx_dw_defer")
        L.check(lib.stlt_ctx_dw_defer(self.context.handle, 1), "stlt_ctx_dw_defer")
        self.context.defer_on = True
        self.context.keep = []
        return self

    def __exit__(self, exc_type, exc, tb):
        lib = L.load()
        c = self.context
        c.defer_on = False
        try:
            if exc_type is None and lib.stlt_ctx_dw_pending(c.handle) > 0:
                dev = torch.device("cuda", torch.cuda.current_device())
                sk = _sk_scratch(dev)
                L.check(lib.stlt_ctx_dw_flush(c.handle, sk.data_ptr(), sk.numel(), _stream()), "stlt_ctx_dw_flush")
        finally:
            lib.stlt_ctx_dw_defer(c.handle, -1)
            c.keep = []  # torch's allocator recycles by the current stream: the flush's launches are enqueued on it
        return False


def _block_buffers(kind: int, rows: int, d: int, device, context: Optional[TrainContext], in_place: bool, keep):
    """(keep, work) buffers of a block backward (include/stlt_hip.h: stlt_block_keep_bytes / _work_bytes).  `work` is the shared per-stream
    buffer.  `keep` holds the gradients the block's weight-gradient products read: the shared per-stream buffer too, unless the call's
    context is deferring them and this block's go in place — then a buffer of its own that lives (with the activations in `keep`) until
    the flush."""
    lib = L.load()
    work = _scratch(int(lib.stlt_block_work_bytes(rows, d)), device, 1)
    nkeep = int(lib.stlt_block_keep_bytes(rows, d, kind))
    if context is not None and context.defer_on:
        if in_place:
            L.check(lib.stlt_ctx_dw_defer(context.handle, 1), "stlt_ctx_dw_defer")
            kb = torch.empty(nkeep, dtype=torch.uint8, device=device)
            context.keep.append((kb,) + tuple(keep))
            return kb, work
        L.check(lib.stlt_ctx_dw_defer(context.handle, 0), "stlt_ctx_dw_defer")  # this block launches its own products (temporaries handed back to autograd)
    return _scratch(nkeep, device, 2), work


def _block_dropout(p: float):
    """(p, seed, site0) of a block call: one seed per call from torch's CPU generator, site ids site0 and site0 + 1."""
    p = float(p)
    if p <= 0.0:
        return 0.0, 0, 0
    # every call draws its own 62-bit seed, so the site ids only have to tell the call's two dropout sites apart: a constant
    # pair keeps a forward repeatable under torch.manual_seed whatever ran before it in the process
    return p, int(torch.randint(0, 2 ** 62, (1,)).item()), 0x400000


def grad_targets(ws, needs):
    """Where a block's backward accumulates its parameter gradients.  Inside a Trainer step (BoundFlatGrads.accumulating) a
    parameter whose .grad is a view of the trainer's flat gradient buffer is accumulated into IN PLACE by the native call (and
    autograd is handed None for it): no zero-filled temporary, no AccumulateGrad add.  Anywhere else — torch.autograd.grad() or a
    hand-written loop on a Trainer-bound model — every parameter gets a fresh zero tensor that autograd returns / accumulates."""
    targets, returned = [], []
    for w, need in zip(ws, needs):
        bound = getattr(w, "_stlt_bound", None)
        view = bound.view_of(w) if (need and bound is not None) else None
        if not need:
            targets.append(None); returned.append(None)
        elif view is not None:
            bound.touch(w)
            targets.append(view); returned.append(None)
        else:
            g = torch.zeros_like(w)
            targets.append(g); returned.append(g)
    return targets, returned


_NO_MASK = {}


def _no_mask(S: int, Lk: int, device) -> torch.Tensor:
    """The all-zeros key-padding mask of a block call without a mask: one cached tensor per (device, shape) instead of a fill kernel per
    call (16 per CACNF step).  Read-only: the kernels never write a mask."""
    key = (device, S, Lk)
    t = _NO_MASK.get(key)
    if t is None:
        if len(_NO_MASK) > 64:
            _NO_MASK.clear()
        _NO_MASK[key] = t = torch.zeros(S, Lk, dtype=torch.uint8, device=device)
    return t


class AttnBlockFn(torch.autograd.Function):
    """One residual attention block as ONE native call each way (include/stlt_hip.h: stlt_attn_block_fwd_train / _bwd_train):
    LN_eps(x + drop(MHA(x, c, c) Woᵀ + bo)) — SelfAttentionLayer / CrossAttentionLayer of the fusion models (models.py:345-382)
    and the attention half of nn.TransformerEncoderLayer.  x (S,Lq,d); c (S,Lk,d) or None (self-attention); kpm (S,Lk) or None."""

    @staticmethod
    def forward(ctx, x, c, kpm, causal, heads, eps, drop_p, in_w, in_b, out_w, out_b, ln_w, ln_b):
        lib = L.load()
        x = _chk(x.contiguous(), torch.float32, "x")
        S, Lq, d = x.shape
        if c is not None:
            c = _chk(c.contiguous(), torch.float32, "c")
        Lk = Lq if c is None else c.shape[1]
        kpm8 = _no_mask(S, Lk, x.device) if kpm is None else _mask_u8(kpm.contiguous(), "kpm")
        q = torch.empty(S * Lq, (3 if c is None else 1) * d, device=x.device, dtype=torch.float32)
        kv = None if c is None else torch.empty(S * Lk, 2 * d, device=x.device, dtype=torch.float32)
        att, a, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        params = L.AttnBlockParams(*[_p(t) for t in (in_w, in_b, out_w, out_b, ln_w, ln_b)])
        p, seed, site = _block_dropout(drop_p)
        sk = _sk_scratch(x.device)
        L.check(lib.stlt_attn_block_fwd_train(C.byref(params), d, heads, float(eps), _p(x), Lq, _p(c), Lk, _p(kpm8), int(bool(causal)), S, p, seed, site,
                                              _p(q), _p(kv), _p(att), _p(a), _p(out), sk.data_ptr(), sk.numel(), _stream()), "stlt_attn_block_fwd_train")
        ctx.save_for_backward(x, c, kpm8, q, kv, att, a)
        ctx.ws = (in_w, in_b, out_w, out_b, ln_w, ln_b)
        ctx.meta = (bool(causal), heads, float(eps), p, seed, site)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, c, kpm8, q, kv, att, a = ctx.saved_tensors
        causal, heads, eps, p, seed, site = ctx.meta
        S, Lq, d = x.shape
        Lk = Lq if c is None else c.shape[1]
        dy = dy.contiguous()
        ws = ctx.ws
        targets, grads = grad_targets(ws, ctx.needs_input_grad[7:13])
        dx = torch.empty_like(x)
        dc = torch.empty_like(c) if (c is not None and ctx.needs_input_grad[1]) else None
        params = L.AttnBlockParams(*[_p(t) for t in ws])
        gstruct = L.AttnBlockParams(*[_p(t) for t in targets])
        in_place = all(r is None for r, need in zip(grads, ctx.needs_input_grad[7:13]) if need)
        tctx = context_of(ws)
        kb, wb = _block_buffers(0, S * max(Lq, Lk), d, x.device, tctx, in_place, (x, c, att, a, q, kv))
        L.check(lib.stlt_attn_block_bwd_train(C.byref(params), C.byref(gstruct), d, heads, eps, _p(x), Lq, _p(c), Lk, _p(kpm8), int(causal), S, p, seed,
                                              site, _p(q), _p(kv), _p(att), _p(a), _p(dy), _p(dx), _p(dc), _ctx_handle(tctx), kb.data_ptr(), kb.numel(),
                                              wb.data_ptr(), wb.numel(), _stream()), "stlt_attn_block_bwd_train")
        return (dx, dc, None, None, None, None, None, *grads)


class FfnBlockFn(torch.autograd.Function):
    """One residual feed-forward block as ONE native call each way: LN_eps(x + drop(W2 drop_inner(act(W1 x + b1)) + b2)) — the
    fusion models' layout_ffn (models.py:384-401; GELU, no inner dropout) and the feed-forward half of nn.TransformerEncoderLayer
    (GELU in the layout towers, ReLU in the appearance encoder; inner dropout)."""

    @staticmethod
    def forward(ctx, x, eps, act, inner_dropout, drop_p, w1, b1, w2, b2, ln_w, ln_b):
        lib = L.load()
        x = _chk(x.contiguous(), torch.float32, "x")
        d = x.shape[-1]
        M = x.numel() // d
        u = torch.empty(M, 4 * d, device=x.device, dtype=torch.float32) if act == L.ACT_GELU else None
        h = torch.empty(M, 4 * d, device=x.device, dtype=torch.float32)
        f, out = torch.empty_like(x), torch.empty_like(x)
        params = L.FfnBlockParams(*[_p(t) for t in (w1, b1, w2, b2, ln_w, ln_b)])
        p, seed, site = _block_dropout(drop_p)
        sk = _sk_scratch(x.device)
        L.check(lib.stlt_ffn_block_fwd_train(C.byref(params), d, float(eps), int(act), int(bool(inner_dropout)), _p(x), M, p, seed, site, _p(u), _p(h),
                                             _p(f), _p(out), sk.data_ptr(), sk.numel(), _stream()), "stlt_ffn_block_fwd_train")
        ctx.save_for_backward(x, u, h, f)
        ctx.ws = (w1, b1, w2, b2, ln_w, ln_b)
        ctx.meta = (float(eps), int(act), bool(inner_dropout), p, seed, site)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, u, h, f = ctx.saved_tensors
        eps, act, inner, p, seed, site = ctx.meta
        d = x.shape[-1]
        M = x.numel() // d
        dy = dy.contiguous()
        ws = ctx.ws
        targets, grads = grad_targets(ws, ctx.needs_input_grad[5:11])
        dx = torch.empty_like(x)
        params = L.FfnBlockParams(*[_p(t) for t in ws])
        gstruct = L.FfnBlockParams(*[_p(t) for t in targets])
        in_place = all(r is None for r, need in zip(grads, ctx.needs_input_grad[5:11]) if need)
        tctx = context_of(ws)
        kb, wb = _block_buffers(1, M, d, x.device, tctx, in_place, (x, u, h, f))
        L.check(lib.stlt_ffn_block_bwd_train(C.byref(params), C.byref(gstruct), d, eps, act, int(inner), _p(x), M, p, seed, site, _p(u), _p(h), _p(f),
                                             _p(dy), _p(dx), _ctx_handle(tctx), kb.data_ptr(), kb.numel(), wb.data_ptr(), wb.numel(), _stream()),
                "stlt_ffn_block_bwd_train")
        return (dx, None, None, None, None, *grads)


class EmbedFn(torch.autograd.Function):
    """K1 (CategoryBoxEmbeddings, models.py:29-39) under autograd: native forward keeping the pre-LayerNorm sum, native backward."""

    @staticmethod
    def forward(ctx, categories, boxes, scores, cat_w, box_w, box_b, score_w, score_b, ln_w, ln_b, eps):
        lib = L.load()
        categories = _chk(categories.contiguous(), torch.int64, "categories")
        boxes = _chk(boxes.contiguous(), torch.float32, "boxes")
        scores = None if scores is None else _chk(scores.contiguous(), torch.float32, "scores")
        tok, d = categories.numel(), cat_w.shape[1]
        pre = torch.empty(*categories.shape, d, device=categories.device, dtype=torch.float32)
        out = torch.empty_like(pre)
        L.check(lib.stlt_embed_fwd_train(_p(categories), _p(boxes), _p(scores), _p(cat_w), cat_w.shape[0], _p(box_w), _p(box_b), _p(score_w),
                                         _p(score_b), _p(ln_w), _p(ln_b), float(eps), tok, d, _p(pre), _p(out), _stream()), "stlt_embed_fwd_train")
        ctx.save_for_backward(categories, boxes, scores, pre, ln_w, cat_w, box_w, score_w)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        categories, boxes, scores, pre, ln_w, cat_w, box_w, score_w = ctx.saved_tensors
        tok, d, Cn = categories.numel(), pre.shape[-1], cat_w.shape[0]
        dy = dy.contiguous()
        d_pre = torch.empty_like(pre)
        g_ln_w, g_ln_b = torch.zeros_like(ln_w), torch.zeros_like(ln_w)
        sc = _scratch(max(int(lib.stlt_add_layernorm_bwd_scratch_bytes(d)), int(lib.stlt_embed_bwd_scratch_bytes(tok, Cn, d))), pre.device)
        L.check(lib.stlt_add_layernorm_bwd(_p(dy), _p(pre), None, _p(ln_w), float(ctx.eps), tok, d, _p(d_pre), _p(g_ln_w), _p(g_ln_b),
                                           sc.data_ptr(), sc.numel(), _stream()), "stlt_add_layernorm_bwd")
        g_cat, g_box_w = torch.zeros_like(cat_w), torch.zeros_like(box_w)
        g_box_b = torch.zeros(d, device=pre.device, dtype=torch.float32)
        g_sw = torch.zeros_like(score_w) if scores is not None else None
        g_sb = torch.zeros(d, device=pre.device, dtype=torch.float32) if scores is not None else None
        L.check(lib.stlt_embed_bwd(_p(d_pre), _p(categories), _p(boxes), _p(scores), Cn, tok, d, _p(g_cat), _p(g_box_w), _p(g_box_b), _p(g_sw),
                                   _p(g_sb), sc.data_ptr(), sc.numel(), _stream()), "stlt_embed_bwd")
        return None, None, None, g_cat, g_box_w, g_box_b, g_sw, g_sb, g_ln_w, g_ln_b, None


class FramesEmbedFn(torch.autograd.Function):
    """K7 (FramesEmbeddings, models.py:98-111) on the (B,T,d) CLS rows under autograd."""

    @staticmethod
    def forward(ctx, cls_rows, frame_types, pos_w, type_w, ln_w, ln_b, eps):
        lib = L.load()
        cls_rows = _chk(cls_rows.contiguous(), torch.float32, "cls_rows")
        frame_types = _chk(frame_types.contiguous(), torch.int64, "frame_types")
        B, T, d = cls_rows.shape
        pre, out = torch.empty_like(cls_rows), torch.empty_like(cls_rows)
        L.check(lib.stlt_frames_embed_fwd_train(_p(cls_rows), d, _p(frame_types), _p(pos_w), _p(type_w), _p(ln_w), _p(ln_b), float(eps), B, T, d,
                                                _p(pre), _p(out), _stream()), "stlt_frames_embed_fwd_train")
        ctx.save_for_backward(frame_types, pre, ln_w, pos_w, type_w)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        frame_types, pre, ln_w, pos_w, type_w = ctx.saved_tensors
        B, T, d = pre.shape
        dy = dy.contiguous()
        d_pre = torch.empty_like(pre)
        g_ln_w, g_ln_b = torch.zeros_like(ln_w), torch.zeros_like(ln_w)
        sc = _scratch(max(int(lib.stlt_add_layernorm_bwd_scratch_bytes(d)), int(lib.stlt_frames_embed_bwd_scratch_bytes(T, d))), pre.device)
        L.check(lib.stlt_add_layernorm_bwd(_p(dy), _p(pre), None, _p(ln_w), float(ctx.eps), B * T, d, _p(d_pre), _p(g_ln_w), _p(g_ln_b),
                                           sc.data_ptr(), sc.numel(), _stream()), "stlt_add_layernorm_bwd")
        g_pos, g_type = torch.zeros_like(pos_w), torch.zeros_like(type_w)
        L.check(lib.stlt_frames_embed_bwd(_p(d_pre), _p(frame_types), B, T, d, _p(g_pos), _p(g_type), sc.data_ptr(), sc.numel(), _stream()),
                "stlt_frames_embed_bwd")
        return d_pre, None, g_pos, g_type, g_ln_w, g_ln_b, None


# ---------------------------------------------------------------------------------------------------------------------
# Device guard: every wrapper above launches on `torch.cuda.current_stream()`, which belongs to the CURRENT device.  A
# caller that holds tensors on another GPU of the same process (cuda:1 while cuda:0 is current) must still get its kernels
# on the tensors' device, so each public wrapper and each autograd Function runs under `torch.cuda.device(<first tensor>)`.
def _guarded(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index == torch.cuda.current_device():
                    break
                with torch.cuda.device(a.device):
                    return fn(*args, **kwargs)
        return fn(*args, **kwargs)

    return wrapper


def _install_device_guards():
    import types
    g = globals()
    for name, obj in list(g.items()):
        if name.startswith("_") or name in ("prof_enable", "prof_collect", "prof_launches", "prof_take_gemm_flops", "workspace_bytes", "dropout"):
            continue
        if isinstance(obj, types.FunctionType) and obj.__module__ == __name__:
            g[name] = _guarded(obj)
        elif isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            obj.forward = staticmethod(_guarded(obj.forward))
            obj.backward = staticmethod(_guarded(obj.backward))


_install_device_guards()
