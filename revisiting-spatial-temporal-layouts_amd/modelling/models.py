"""Drop-in ``StltBackbone`` / ``Stlt`` for the reference's ``src/modelling/models.py:16-195``.

Same constructor (``StltModelConfig``), same ``forward(batch: Dict[str, Tensor])`` signature and return
values, same 174 state-dict keys (168 for the backbone) — but the modules below only *store* parameters
(``nn.Parameter`` holders with torch's default initialisation).  All arithmetic runs in hand-written HIP
kernels for gfx950 through the C-ABI of ``libstlt_hip.so``; no ``torch.nn`` forward is ever called and
there is no CPU fallback (CPU tensors raise ``StltHipError``).
"""
from __future__ import annotations

import copy
import ctypes as C
import math
from typing import Dict, List, Optional

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from .configs import StltModelConfig, model_configs_factory  # noqa: F401  (re-exported like the reference)

_ENC_EPS = 1e-5  # nn.TransformerEncoderLayer default; config.layer_norm_eps is not forwarded (models.py:46-52,118-124)


class _SelfAttnParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names/shapes/init: in_proj_weight (3d,d) rows [q;k;v]."""

    def __init__(self, d: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _EncoderLayerParams(nn.Module):
    """Parameter holder with nn.TransformerEncoderLayer's 12 tensors (post-norm, gelu, dim_feedforward=4d)."""

    def __init__(self, d: int):
        super().__init__()
        self.self_attn = _SelfAttnParams(d)
        self.linear1 = nn.Linear(d, 4 * d)
        self.linear2 = nn.Linear(4 * d, d)
        self.norm1 = nn.LayerNorm(d, eps=_ENC_EPS)
        self.norm2 = nn.LayerNorm(d, eps=_ENC_EPS)

    def c_struct(self) -> L.LayerParams:
        sa = self.self_attn
        ts = (sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, self.linear1.weight,
              self.linear1.bias, self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias,
              self.norm2.weight, self.norm2.bias)
        return L.LayerParams(*[_dev_ptr(t) for t in ts])


class _EncoderStack(nn.Module):
    """``nn.TransformerEncoder``'s container: ``layers`` are deep copies of one layer (same initial weights)."""

    def __init__(self, layer: _EncoderLayerParams, num_layers: int):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])


def _dev_ptr(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise L.StltHipError("STLT parameters must be on a GPU (`model.to('cuda')`): the hot path has no CPU fallback")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise L.StltHipError("STLT parameters must be contiguous float32")
    return t.data_ptr()


class CategoryBoxEmbeddings(nn.Module):
    """Parameters of reference models.py:16-27; forward = ``stlt_embed_fwd`` (K1)."""

    def __init__(self, config: StltModelConfig):
        super().__init__()
        self.category_embeddings = nn.Embedding(config.unique_categories, config.hidden_size, padding_idx=0)
        self.box_embedding = nn.Linear(4, config.hidden_size)
        self.score_embeddings = nn.Linear(1, config.hidden_size)
        self.layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.eps = config.layer_norm_eps

    def forward(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        scores = batch["scores"].contiguous() if "scores" in batch else None
        return ops.embed(batch["categories"].contiguous(), batch["boxes"].contiguous(), scores,
                         self.category_embeddings.weight, self.box_embedding.weight, self.box_embedding.bias,
                         self.score_embeddings.weight, self.score_embeddings.bias, self.layer_norm.weight,
                         self.layer_norm.bias, self.eps)


class SpatialTransformer(nn.Module):
    """Parameters of reference models.py:42-55 (incl. the never-used ``encoder_layer`` the state dict carries)."""

    def __init__(self, config: StltModelConfig):
        super().__init__()
        self.category_box_embeddings = CategoryBoxEmbeddings(config)
        self.encoder_layer = _EncoderLayerParams(config.hidden_size)
        self.transformer = _EncoderStack(self.encoder_layer, config.num_spatial_layers)


class FramesEmbeddings(nn.Module):
    """Parameters of reference models.py:84-96."""

    def __init__(self, config: StltModelConfig):
        super().__init__()
        self.layout_embedding = SpatialTransformer(config)
        self.position_embeddings = nn.Embedding(config.layout_num_frames, config.hidden_size)
        self.frame_type_embedding = nn.Embedding(5, config.hidden_size, padding_idx=0)
        self.layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(config.layout_num_frames).expand((1, -1)))


class ClassificationHead(nn.Module):
    """Parameters of reference models.py:155-160; forward = fc2(LN(gelu(fc1(h)))) on the HIP kernels."""

    def __init__(self, config: StltModelConfig):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.hidden_size)
        self.layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.fc2 = nn.Linear(config.hidden_size, config.num_classes)
        self.eps = config.layer_norm_eps

    def forward(self, hidden_state: torch.Tensor) -> torch.Tensor:
        h = ops.linear(hidden_state.contiguous(), self.fc1.weight, self.fc1.bias, act=L.ACT_GELU)
        h = ops.add_layernorm(h, None, self.layer_norm.weight, self.layer_norm.bias, self.eps)
        return ops.linear(h, self.fc2.weight, self.fc2.bias)


class _Workspace:
    """Grow-only scratch buffer per device, reused across forwards (the C-ABI never allocates)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.device != device or self.buf.numel() < nbytes:
            self.buf = None
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf


def _prep_inputs(batch: Dict[str, torch.Tensor], need_lengths: bool):
    """Validate the collated batch (reference src/modelling/datasets.py:243-288) and build the C struct."""
    cats = batch["categories"]
    if cats.dim() != 3:
        raise L.StltHipError(f"categories must be (B,T,N), got {tuple(cats.shape)}")
    B, T, N = cats.shape
    keep = []  # keep converted tensors alive until the launch is enqueued

    def take(t, dtype, name, shape):
        if tuple(t.shape) != shape:
            raise L.StltHipError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        if dtype == torch.uint8:
            t = ops._mask_u8(t, name)
        else:
            t = ops._chk(t.contiguous(), dtype, name)
        keep.append(t)
        return t.data_ptr()

    inp = L.Inputs()
    inp.B, inp.T, inp.N = B, T, N
    inp.categories = take(cats, torch.int64, "categories", (B, T, N))
    inp.boxes = take(batch["boxes"], torch.float32, "boxes", (B, T, N, 4))
    inp.scores = take(batch["scores"], torch.float32, "scores", (B, T, N)) if "scores" in batch else None
    inp.kpm_boxes = take(batch["src_key_padding_mask_boxes"], torch.uint8, "src_key_padding_mask_boxes", (B, T, N))
    inp.frame_types = take(batch["frame_types"], torch.int64, "frame_types", (B, T))
    inp.kpm_frames = take(batch["src_key_padding_mask_frames"], torch.uint8, "src_key_padding_mask_frames", (B, T))
    inp.lengths = take(batch["lengths"], torch.int64, "lengths", (B,)) if need_lengths else None
    # optional: the batch's real rows, counted where the masks were made (collate.real_counts on the host).  With them a skip-padding
    # forward / training step reads nothing back from the device (include/stlt_hip.h: stlt_inputs.n_real_tokens)
    if "num_real_tokens" in batch or "num_real_frames" in batch:
        counts = []
        for key in ("num_real_tokens", "num_real_frames"):
            v = batch.get(key)
            if isinstance(v, torch.Tensor):
                if v.is_cuda:
                    raise L.StltHipError(f"{key} must be a host integer (a device tensor would have to be read back, which is what it is there to avoid)")
                v = int(v)
            if not isinstance(v, int) or v <= 0:
                raise L.StltHipError("num_real_tokens and num_real_frames come together, as positive host integers")
            counts.append(v)
        inp.n_real_tokens, inp.n_real_frames = counts
    return inp, keep, (B, T, N)


class StltBackbone(nn.Module):
    """Drop-in for reference ``StltBackbone`` (models.py:114-152): ``forward(batch) -> (T, B, d)``."""

    def __init__(self, config: StltModelConfig):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self.config = config
        self.frames_embeddings = FramesEmbeddings(config)
        self.transformer = _EncoderStack(_EncoderLayerParams(config.hidden_size), config.num_temporal_layers)
        self.cls_only_last_spatial = True  # exact: only token 0 of the last spatial layer is read (models.py:79)
        self.last_row_only_temporal = True  # exact, Stlt.forward only: the head reads one row per clip (models.py:189-192)
        # Stlt.forward (inference and training), off by default: compute the real tokens / frames of the padded batch only.
        # Same logits (a padded row is masked as a key and never read); needs collater-shaped masks and costs one
        # stream synchronisation per call (include/stlt_hip.h: STLT_FLAG_SKIP_PADDING).
        self.skip_padding = False
        self._cache = None
        self._ws = _Workspace()

    def __getstate__(self):  # ctypes tables / scratch are rebuilt lazily; keep deepcopy / pickling working
        state = self.__dict__.copy()
        state["_cache"] = None
        state["_ws"] = _Workspace()
        state.pop("_train_bufs", None)
        state.pop("_flat_grad_buf", None)
        state.pop("_grad_table", None)
        return state

    @classmethod
    def from_pretrained(cls, config: StltModelConfig):
        model = cls(config)
        model.load_state_dict(torch.load(config.load_backbone_path, map_location="cpu"))
        return model

    # ---- parameter table for the C-ABI -------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        self._cache = None  # .to()/.cuda()/.float() move parameter storage
        self.__dict__.pop("_grad_table", None)
        return super()._apply(fn, *a, **kw)

    def _sentinel(self):
        """Identity of every parameter's storage: the cached C table holds raw device pointers, so rebinding any
        parameter (`p.data = …`, an EMA / SWA swap, pruning, replacing a Parameter) must rebuild it."""
        return tuple(q.data_ptr() for q in self.parameters())

    def _build_struct(self, head: Optional["ClassificationHead"], ptr):
        """Fill a stlt_params table; `ptr(tensor)` yields the device pointer to store for that parameter (its data
        for the forward tables, its gradient buffer — or None — for the backward's gradient table)."""
        cfg = self.config
        fe = self.frames_embeddings
        le = fe.layout_embedding
        cbe = le.category_box_embeddings

        def layer_struct(l):
            sa = l.self_attn
            ts = (sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, l.linear1.weight,
                  l.linear1.bias, l.linear2.weight, l.linear2.bias, l.norm1.weight, l.norm1.bias, l.norm2.weight,
                  l.norm2.bias)
            return L.LayerParams(*[ptr(t) for t in ts])

        sp = (L.LayerParams * max(1, len(le.transformer.layers)))(*[layer_struct(l) for l in le.transformer.layers])
        tp = (L.LayerParams * max(1, len(self.transformer.layers)))(*[layer_struct(l) for l in self.transformer.layers])
        p = L.Params()
        p.d, p.H = cfg.hidden_size, cfg.num_attention_heads
        p.n_categories = cbe.category_embeddings.weight.shape[0]
        p.n_spatial, p.n_temporal = len(le.transformer.layers), len(self.transformer.layers)
        p.n_classes = 0 if head is None else head.fc2.weight.shape[0]
        p.n_positions = fe.position_embeddings.weight.shape[0]
        p.ln_eps = cfg.layer_norm_eps
        for name, t in (("cat_emb", cbe.category_embeddings.weight), ("box_w", cbe.box_embedding.weight),
                        ("box_b", cbe.box_embedding.bias), ("score_w", cbe.score_embeddings.weight),
                        ("score_b", cbe.score_embeddings.bias), ("emb_ln_w", cbe.layer_norm.weight),
                        ("emb_ln_b", cbe.layer_norm.bias), ("pos_emb", fe.position_embeddings.weight),
                        ("type_emb", fe.frame_type_embedding.weight), ("frames_ln_w", fe.layer_norm.weight),
                        ("frames_ln_b", fe.layer_norm.bias)):
            setattr(p, name, ptr(t))
        p.spatial, p.temporal = sp, tp
        if head is not None:
            for name, t in (("fc1_w", head.fc1.weight), ("fc1_b", head.fc1.bias), ("head_ln_w", head.layer_norm.weight),
                            ("head_ln_b", head.layer_norm.bias), ("fc2_w", head.fc2.weight), ("fc2_b", head.fc2.bias)):
                setattr(p, name, ptr(t))
        return p, sp, tp  # keep the layer arrays alive with the struct

    def c_params(self, head: Optional["ClassificationHead"] = None):
        key = (self._sentinel(), None if head is None else tuple(q.data_ptr() for q in head.parameters()))
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        self._cache = (key, self._build_struct(head, _dev_ptr))
        return self._cache[1]

    def _train_buf(self, name: str, nbytes: int, device) -> torch.Tensor:
        """Tape / scratch of the training step, allocated zero-filled and reused while the byte size matches.  Two batch
        shapes can round to the same size with different row layouts; the library does not rely on what an earlier step
        left behind: every step it clears the rows its weight-gradient products read beyond the row count (train.hip)."""
        bufs = self.__dict__.setdefault("_train_bufs", {})
        cur = bufs.get(name)
        if cur is None or cur.device != device or cur.numel() != nbytes:
            bufs[name] = cur = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        return cur

    def _flags(self) -> int:
        return ((L.FLAG_CLS_ONLY_LAST_SPATIAL if self.cls_only_last_spatial else 0)
                | (L.FLAG_LAST_ROW_ONLY_TEMPORAL if self.last_row_only_temporal else 0)
                | (L.FLAG_SKIP_PADDING if self.skip_padding else 0))

    def _dropout_live(self) -> bool:
        """nn.Dropout is active whenever the module is in training mode, grad or no grad (models.py:27,37,93,109): such
        forwards take the training kernels (counter-based masks), not the inference schedule."""
        return self.training and self.config.hidden_dropout_prob > 0

    # ---- differentiable forward, composed from the op-level autograd Functions of ops.py ----------------------------
    def _encoder_layer_train(self, l: _EncoderLayerParams, x: torch.Tensor, kpm, causal: bool) -> torch.Tensor:
        """nn.TransformerEncoderLayer as configured at models.py:46-52,118-124 (post-norm, GELU, eps 1e-5) as two native
        block calls each way (attention half, feed-forward half; ops.AttnBlockFn / ops.FfnBlockFn), dropout with the native
        counter-based masks at the reference's four sites."""
        p, H = self.config.hidden_dropout_prob if self.training else 0.0, self.config.num_attention_heads
        sa = l.self_attn
        x = ops.AttnBlockFn.apply(x, None, kpm, causal, H, _ENC_EPS, p, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                                  l.norm1.weight, l.norm1.bias)
        return ops.FfnBlockFn.apply(x, _ENC_EPS, L.ACT_GELU, True, p, l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias,
                                    l.norm2.weight, l.norm2.bias)

    def forward_train(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """(B,T,d) backbone output with an autograd graph: what a model that consumes EVERY row of the backbone (the fusion
        models) trains through.  `Stlt` itself trains through the single native reverse sweep instead (`_StltTrainFn`).
        Layouts of at most 256 frames / 256 object slots (the op-level attention backward streams keys above 64)."""
        if not self.skip_padding and "lengths" in batch:
            # one native tape forward / reverse sweep (csrc/train.hip with STLT_FLAG_TRAIN_BACKBONE): the last spatial layer
            # runs its out-proj / norms / FFN on the CLS rows only, weight gradients go out layer by layer in grouped launches
            return _BackboneTrainFn.apply(self, batch, *tuple(self.parameters()))
        fe = self.frames_embeddings
        le = fe.layout_embedding
        cbe = le.category_box_embeddings
        cats = batch["categories"]
        B, T, N = cats.shape
        eps, p = self.config.layer_norm_eps, self.config.hidden_dropout_prob
        x = ops.EmbedFn.apply(cats, batch["boxes"], batch.get("scores"), cbe.category_embeddings.weight, cbe.box_embedding.weight,
                              cbe.box_embedding.bias, cbe.score_embeddings.weight, cbe.score_embeddings.bias, cbe.layer_norm.weight,
                              cbe.layer_norm.bias, eps)
        x = ops.dropout(x, p, self.training).view(B * T, N, -1)
        kpm_boxes = batch["src_key_padding_mask_boxes"].reshape(B * T, N)
        for l in le.transformer.layers:
            x = self._encoder_layer_train(l, x, kpm_boxes, False)
        cls_rows = x[:, 0].reshape(B, T, -1)  # models.py:79
        g = ops.FramesEmbedFn.apply(cls_rows, batch["frame_types"], fe.position_embeddings.weight, fe.frame_type_embedding.weight,
                                    fe.layer_norm.weight, fe.layer_norm.bias, eps)
        g = ops.dropout(g, p, self.training)
        kpm_frames = batch["src_key_padding_mask_frames"]
        for l in self.transformer.layers:
            g = self._encoder_layer_train(l, g, kpm_frames, True)
        return g

    def forward_batch_major(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """HIP forward, batch-major (B,T,d) result (the layout the kernels compute in)."""
        if (torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())) or self._dropout_live():
            return self.forward_train(batch)  # under no_grad the op-level Functions just run their forward kernels
        lib = L.load()
        inp, keep, (B, T, N) = _prep_inputs(batch, need_lengths=False)
        device = batch["categories"].device
        p, _, _ = self.c_params()
        d = self.config.hidden_size
        nbytes = ops.workspace_bytes(B, T, N, d, 0)
        ws = self._ws.get(nbytes, device)
        out = torch.empty(B, T, d, device=device, dtype=torch.float32)
        with torch.cuda.device(device):
            L.check(lib.stlt_backbone_forward(C.byref(p), C.byref(inp), ws.data_ptr(), ws.numel(), self._flags(),
                                              out.data_ptr(), torch.cuda.current_stream().cuda_stream),
                    "stlt_backbone_forward")
        return out

    def forward(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        # [Num. frames, Batch size, Hidden size] like the reference (a transposed view of the batch-major result)
        return self.forward_batch_major(batch).transpose(0, 1)


class Stlt(nn.Module):
    """Drop-in for reference ``Stlt`` (models.py:166-195): ``forward(batch) -> {"stlt": (B, num_classes)}``."""

    def __init__(self, config: StltModelConfig):
        super().__init__()
        self.config = config
        if config.load_backbone_path is not None:
            self.backbone = StltBackbone.from_pretrained(config)
            if config.freeze_backbone:
                for param in self.backbone.parameters():
                    param.requires_grad = False
        else:
            self.backbone = StltBackbone(config)
        self.prediction_head = ClassificationHead(config)
        self.logit_names = ("stlt",)

    def _own_context(self):
        """The module's own training context (include/stlt_hip.h: stlt_ctx), made at the first autograd backward that runs outside a
        Trainer step: it owns the side stream of that sweep's weight-gradient products.  Not a parameter, not in the state dict."""
        c = self.__dict__.get("_ctx_own")
        if c is None or c.handle is None:
            c = self.__dict__["_ctx_own"] = ops.TrainContext()
        return c

    def train(self, mode: bool = True):
        super().train(mode)
        if self.config.load_backbone_path and self.config.freeze_backbone:
            self.backbone.train(False)
        return self  # the reference returns None here; returning self is a superset

    def _grad_params(self, has_scores: bool):
        """ids of the parameters the forward actually uses (the others get no gradient, like in the reference: the
        dead `encoder_layer` copy, and `score_embeddings` when the batch has no scores — SURVEY.md §5)."""
        le = self.backbone.frames_embeddings.layout_embedding
        skip = {id(q) for q in le.encoder_layer.parameters()}
        if not has_scores:
            skip |= {id(q) for q in le.category_box_embeddings.score_embeddings.parameters()}
        return {id(q) for q in self.parameters()} - skip

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        bb = self.backbone
        grad_path = torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())
        if grad_path or bb._dropout_live():
            # train mode without grad (a validation pass that forgot model.train(False), MC-dropout): the reference still
            # applies its dropouts, so the training forward runs here too — same kernels, the tape is just not kept
            params = tuple(self.parameters())
            logits = _StltTrainFn.apply(self, batch, *params)
            return {k: v for k, v in zip(self.logit_names, (logits,))}
        lib = L.load()
        inp, keep, (B, T, N) = _prep_inputs(batch, need_lengths=True)
        device = batch["categories"].device
        p, _, _ = bb.c_params(self.prediction_head)
        d, K = self.config.hidden_size, self.prediction_head.fc2.weight.shape[0]
        nbytes = ops.workspace_bytes(B, T, N, d, K)
        ws = bb._ws.get(nbytes, device)
        logits = torch.empty(B, K, device=device, dtype=torch.float32)
        with torch.cuda.device(device):
            L.check(lib.stlt_forward(C.byref(p), C.byref(inp), ws.data_ptr(), ws.numel(), bb._flags(), None,
                                     logits.data_ptr(), torch.cuda.current_stream().cuda_stream), "stlt_forward")
        return {k: v for k, v in zip(self.logit_names, (logits,))}


class _StltTrainFn(torch.autograd.Function):
    """Autograd shell of the native training step: forward = stlt_train_forward (records the tape), backward =
    stlt_train_backward (the reverse sweep in HIP).  The parameters are passed as inputs only so that autograd
    routes their gradients; all arithmetic happens behind the C-ABI."""

    @staticmethod
    def forward(ctx, model, batch, *params):
        lib = L.load()
        bb = model.backbone
        inp, keep, (B, T, N) = _prep_inputs(batch, need_lengths=True)
        device = batch["categories"].device
        p, _, _ = bb.c_params(model.prediction_head)
        cfg = model.config
        d, K = cfg.hidden_size, model.prediction_head.fc2.weight.shape[0]
        n_sp, n_tp = p.n_spatial, p.n_temporal
        tape = bb._train_buf("tape", int(lib.stlt_train_tape_bytes(B, T, N, d, n_sp, n_tp)), device)
        # one tape per backbone: a second grad-enabled forward overwrites it, so every forward takes a new generation
        # number and the backward refuses to run on a tape that is no longer its own
        bb._tape_gen = ctx.tape_gen = getattr(bb, "_tape_gen", 0) + 1
        logits = torch.empty(B, K, device=device, dtype=torch.float32)
        # train-mode dropout (reference default hidden_dropout_prob = 0.1): counter-based masks from one seed per
        # forward, drawn from torch's CPU generator (so torch.manual_seed makes runs repeatable)
        drop_p = float(cfg.hidden_dropout_prob) if bb.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop_p > 0 else 0
        seed = getattr(model, "_dropout_seed_override", None) or seed
        flags = L.FLAG_SKIP_PADDING if bb.skip_padding else 0  # the other two flags are inference-only elisions
        with torch.cuda.device(device):
            L.check(lib.stlt_train_forward(C.byref(p), C.byref(inp), tape.data_ptr(), tape.numel(), logits.data_ptr(),
                                           drop_p, seed, flags, torch.cuda.current_stream().cuda_stream), "stlt_train_forward")
        ctx.model, ctx.batch, ctx.shape, ctx.params, ctx.drop = model, batch, (B, T, N, d), params, (drop_p, seed, flags)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = L.load()
        model, batch = ctx.model, ctx.batch
        bb = model.backbone
        B, T, N, d = ctx.shape
        device = dlogits.device
        if getattr(bb, "_tape_gen", 0) != ctx.tape_gen:
            raise L.StltHipError("Stlt backward: the activation tape was overwritten by a later grad-enabled forward of the same "
                                 "model (one tape per backbone): run each forward's backward before the next forward, or wrap "
                                 "forwards that need no gradient in torch.no_grad()")
        inp, keep, _ = _prep_inputs(batch, need_lengths=True)
        p, _, _ = bb.c_params(model.prediction_head)
        used = model._grad_params("scores" in batch)
        want = [prm for prm in ctx.params if prm.requires_grad and id(prm) in used]
        # every gradient lives in one flat buffer (offsets rounded up to 4 floats so each tensor stays 16-byte aligned;
        # the gaps stay zero): what a data-parallel run all-reduces and what train.FusedAdamW consumes
        layout, off = [], 0
        for q in want:
            layout.append((q, off, q.numel()))
            off += (q.numel() + 3) // 4 * 4
        # The fused trainer consumes the buffer before the next backward, so it is kept per backbone and cleared, not
        # re-allocated (344 MB at d = 768).  Otherwise the views below become the parameters' .grad tensors and outlive
        # this call: a fresh buffer every time.
        reuse = bool(getattr(model, "_flat_grads_only", False))
        flat = bb.__dict__.get("_flat_grad_buf") if reuse else None
        if flat is not None and flat.device == device and flat.numel() == off:
            flat.zero_()
        else:
            flat = torch.zeros(off, device=device, dtype=torch.float32)
            if reuse:
                bb.__dict__["_flat_grad_buf"] = flat
        views = {id(q): flat[o: o + n].view_as(q) for q, o, n in layout}
        g, gsp, gtp = bb._build_struct(model.prediction_head, lambda t: views[id(t)].data_ptr() if id(t) in views else None)
        tape = bb._train_buf("tape", int(lib.stlt_train_tape_bytes(B, T, N, d, p.n_spatial, p.n_temporal)), device)
        scratch = bb._train_buf("scratch", int(lib.stlt_train_scratch_bytes(B, T, N, d, p.n_categories)), device)
        dl = dlogits.contiguous().float()
        model._last_flat_grad = flat
        model._flat_layout = layout

        # the training context the sweep names: the trainer's while its step runs this backward (its transposed weight copies, its side
        # stream), else one the module owns for the side stream of plain autograd backwards
        tctx = getattr(model, "_train_context", None) or model._own_context()

        def run(extra_flags):
            L.check(lib.stlt_train_backward(C.byref(p), C.byref(g), C.byref(inp), tape.data_ptr(), tape.numel(),
                                            scratch.data_ptr(), scratch.numel(), dl.data_ptr(), ctx.drop[0], ctx.drop[1],
                                            ctx.drop[2] | extra_flags, tctx.handle, torch.cuda.current_stream().cuda_stream), "stlt_train_backward")

        sync = getattr(model, "_grad_sync", None)  # data-parallel hook: sync(flat, lo, hi) may start reducing flat[lo:hi]
        with torch.cuda.device(device):
            if sync is None:
                run(0)
            else:
                # the temporal tower and the head come last in parameter order and first in the reverse sweep: their
                # gradients are final after the upper half, and can travel while the lower half computes
                upper = {id(q) for q in model.backbone.transformer.parameters()} | {id(q) for q in model.prediction_head.parameters()}
                split = min((o for q, o, n in layout if id(q) in upper), default=off)
                assert all((id(q) in upper) == (o >= split) for q, o, n in layout), "temporal tower + head must be the tail of the flat buffer"
                run(L.FLAG_TRAIN_UPPER_ONLY)
                sync(flat, split, off)
                run(L.FLAG_TRAIN_LOWER_ONLY)
                sync(flat, 0, split)
        model._last_flat_grad = flat  # one contiguous buffer: what a data-parallel wrapper all-reduces
        model._flat_layout = layout
        if getattr(model, "_flat_grads_only", False):  # train.FusedAdamW reads the flat buffer: skip the per-parameter .grad copies
            return (None, None) + tuple(None for _ in ctx.params)
        return (None, None) + tuple(views.get(id(prm)) for prm in ctx.params)


class _BackboneTrainFn(torch.autograd.Function):
    """The backbone alone under autograd, for models that consume EVERY row of its output (the fusion models' layout branch,
    models.py:446-483): the same native tape forward / reverse sweep as `_StltTrainFn` with STLT_FLAG_TRAIN_BACKBONE — no
    prediction head, every temporal layer on every frame, the (B,T,d) output's gradient as the sweep's seed."""

    @staticmethod
    def forward(ctx, bb, batch, *params):
        lib = L.load()
        inp, keep, (B, T, N) = _prep_inputs(batch, need_lengths=True)
        device = batch["categories"].device
        p, _, _ = bb.c_params(None)
        cfg = bb.config
        d = cfg.hidden_size
        tape = bb._train_buf("tape", int(lib.stlt_train_tape_bytes(B, T, N, d, p.n_spatial, p.n_temporal)), device)
        bb._tape_gen = ctx.tape_gen = getattr(bb, "_tape_gen", 0) + 1
        out = torch.empty(B, T, d, device=device, dtype=torch.float32)
        drop_p = float(cfg.hidden_dropout_prob) if bb.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop_p > 0 else 0
        with torch.cuda.device(device):
            L.check(lib.stlt_train_forward(C.byref(p), C.byref(inp), tape.data_ptr(), tape.numel(), out.data_ptr(), drop_p, seed,
                                           L.FLAG_TRAIN_BACKBONE, torch.cuda.current_stream().cuda_stream), "stlt_train_forward")
        ctx.bb, ctx.batch, ctx.shape, ctx.params, ctx.drop = bb, batch, (B, T, N, d), params, (drop_p, seed)
        ctx.inp = (inp, keep)  # the input table and the tensors it points into, for the backward
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        bb, batch = ctx.bb, ctx.batch
        B, T, N, d = ctx.shape
        device = dout.device
        if getattr(bb, "_tape_gen", 0) != ctx.tape_gen:
            raise L.StltHipError("StltBackbone backward: the activation tape was overwritten by a later grad-enabled forward of the same "
                                 "backbone (one tape per backbone): run each forward's backward before the next forward")
        inp, keep = ctx.inp
        p, _, _ = bb.c_params(None)
        # inside a Trainer step, parameters whose .grad is bound to the trainer's flat buffer are accumulated into in place (no
        # temporary, autograd gets None for them); the others — and every parameter outside a Trainer step, e.g. under
        # torch.autograd.grad() — get views of a fresh zero buffer that autograd accumulates / returns.  The gradient table of a
        # fully bound backbone is the same every step (fixed views of one flat buffer): it is kept with the backbone.
        b0 = next((getattr(q, "_stlt_bound", None) for q in ctx.params if q.requires_grad), None)
        key = (id(b0), "scores" in batch, tuple(q.requires_grad for q in ctx.params)) if (b0 is not None and b0.accumulating) else None
        cached = bb.__dict__.get("_grad_table")
        views = {}
        if key is not None and cached is not None and cached[0] == key and cached[1] is b0:
            g, gsp, gtp, touched = cached[2]
            b0._touched.update(touched)
        else:
            le = bb.frames_embeddings.layout_embedding
            skip = {id(q) for q in le.encoder_layer.parameters()}
            if "scores" not in batch:
                skip |= {id(q) for q in le.category_box_embeddings.score_embeddings.parameters()}
            want = [prm for prm in ctx.params if prm.requires_grad and id(prm) not in skip]
            direct = {}
            for q in want:
                bound = getattr(q, "_stlt_bound", None)
                view = bound.view_of(q) if bound is not None else None
                if view is not None:
                    bound.touch(q)
                    direct[id(q)] = view
            layout, off = [], 0
            for q in want:
                if id(q) not in direct:
                    layout.append((q, off, q.numel()))
                    off += (q.numel() + 3) // 4 * 4
            flat = torch.zeros(off, device=device, dtype=torch.float32) if off else None
            views = {id(q): flat[o: o + n].view_as(q) for q, o, n in layout}
            target = lambda t: direct[id(t)].data_ptr() if id(t) in direct else (views[id(t)].data_ptr() if id(t) in views else None)  # noqa: E731
            g, gsp, gtp = bb._build_struct(None, target)
            if key is not None and not layout and all(getattr(q, "_stlt_bound", None) is b0 for q in want):
                bb.__dict__["_grad_table"] = (key, b0, (g, gsp, gtp, frozenset(direct)))
        tape = bb._train_buf("tape", int(lib.stlt_train_tape_bytes(B, T, N, d, p.n_spatial, p.n_temporal)), device)
        scratch = bb._train_buf("scratch", int(lib.stlt_train_scratch_bytes(B, T, N, d, p.n_categories)), device)
        dl = dout.contiguous().float()
        with torch.cuda.device(device):
            L.check(lib.stlt_train_backward(C.byref(p), C.byref(g), C.byref(inp), tape.data_ptr(), tape.numel(), scratch.data_ptr(), scratch.numel(),
                                            dl.data_ptr(), ctx.drop[0], ctx.drop[1], L.FLAG_TRAIN_BACKBONE, ops._ctx_handle(ops.context_of(ctx.params)),
                                            torch.cuda.current_stream().cuda_stream), "stlt_train_backward")
        return (None, None) + tuple(views.get(id(prm)) for prm in ctx.params)


models_factory = {"stlt": Stlt}  # "caf" / "cacnf" are added by modelling/fusion.py at package import
