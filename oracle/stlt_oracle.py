"""CPU oracle for the STLT forward hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product path (``revisiting-spatial-temporal-
layouts_amd``) never does: it runs hand-written HIP through the C-ABI and fails
loudly when the extension is missing.

This is a plain-tensor restatement (batch-major, no ``nn.Module``) of the
reference's algorithm.  The reference is pure Python on ``torch.nn`` (pinned
torch 1.10.1, ``poetry.lock:938-939``); all arithmetic lives in torch, so the
restatement is written from the published semantics of ``nn.Embedding``,
``nn.Linear``, ``nn.LayerNorm``, ``F.gelu`` (exact erf) and
``nn.TransformerEncoderLayer`` (post-norm, ``F.multi_head_attention_forward``)
and anchored on the reference call sites cited per function.

PARITY PIN: the reference holds no tests, golden vectors or fixtures for this
path.  The pin is therefore the reference itself, imported
in the build container by ``tools/gen_golden.py`` (torch 2.10.0 CPU fp32):
its logits / intermediates on seeded inputs + closed-form weights are committed
under ``tests/golden/`` and ``tests/test_oracle_golden.py`` checks this oracle
against them (fp32 <= 2e-5 abs on logits, fp64-run <= 5e-6).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

_LAYER_KEYS = (
    "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
    "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
    "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias",
)


# ---- train-mode dropout: the build's counter-based masks restated (include/stlt_hip.h, "training step") -------------
# The reference uses nn.Dropout at six sites; its Philox stream cannot be reproduced, so the mask
# DEFINITION is the build's own (csrc/common.h: stlt_keep): key = splitmix64-finaliser(seed*G1 + site*G2); keep iff
# mix32(idx; key) >= p*2^32 with a keyed two-round 32-bit multiply-xorshift (cheap enough for a GEMM epilogue).
_M64 = (1 << 64) - 1
_M32 = (1 << 32) - 1


def dropout_key(seed: int, site: int) -> int:
    z = (seed * 0x9E3779B97F4A7C15 + site * 0xD1B54A32D192ED03) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def dropout_keep(p: float, seed: int, site: int, idx: "np.ndarray"):
    import numpy as np
    key = dropout_key(seed, site)
    k0, k1 = np.uint32(key & _M32), np.uint32(key >> 32)
    idx = idx.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (idx & np.uint64(_M32)).astype(np.uint32) + k0 + (idx >> np.uint64(32)).astype(np.uint32) * np.uint32(0x9E3779B9)
        x = (x ^ (x >> np.uint32(16))) * np.uint32(0x7FEB352D)
        x = x ^ k1
        x = (x ^ (x >> np.uint32(15))) * np.uint32(0x846CA68B)
        x = x ^ (x >> np.uint32(16))
    thr = min(int(p * 4294967296.0), 4294967295)
    return x >= np.uint32(thr)


class Dropout:
    """p, seed -> multiplicative masks (kept entries carry float32(1/(1-p)), as the kernels multiply by it)."""

    def __init__(self, p: float, seed: int):
        self.p, self.seed = float(p), int(seed)
        import numpy as np
        self.scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))

    def elementwise(self, site: int, x: torch.Tensor) -> torch.Tensor:
        import numpy as np
        if self.p <= 0:
            return x
        idx = np.arange(x.numel(), dtype=np.uint64)
        keep = dropout_keep(self.p, self.seed, site, idx).reshape(tuple(x.shape))
        return x * (torch.from_numpy(keep).to(x.dtype) * self.scale)

    def attention(self, site: int, probs: torch.Tensor) -> torch.Tensor:
        """probs (S,H,L,L): element (s,h,i,j) has idx = (((s*L+i)*H + h) << 8) | j."""
        import numpy as np
        if self.p <= 0:
            return probs
        S, H, L, _ = probs.shape
        s_, h_, i_, j_ = np.meshgrid(np.arange(S, dtype=np.uint64), np.arange(H, dtype=np.uint64),
                                     np.arange(L, dtype=np.uint64), np.arange(L, dtype=np.uint64), indexing="ij")
        idx = ((((s_ * np.uint64(L) + i_) * np.uint64(H)) + h_) << np.uint64(8)) | j_
        keep = dropout_keep(self.p, self.seed, site, idx)
        return probs * (torch.from_numpy(keep).to(probs.dtype) * self.scale)


SITE_EMBED, SITE_FRAMES = 0xE0, 0xE1


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """nn.LayerNorm over the last dim: biased variance, eps inside the sqrt."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def gelu(x: torch.Tensor) -> torch.Tensor:
    """Exact erf GELU (``activation="gelu"`` models.py:51,123; ``F.gelu`` models.py:163)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def category_box_embeddings(sd: Dict[str, torch.Tensor], prefix: str, batch: Dict[str, torch.Tensor],
                            eps: float) -> torch.Tensor:
    """CategoryBoxEmbeddings.forward — src/modelling/models.py:29-39. -> (B,T,N,d)"""
    E = sd[prefix + "category_embeddings.weight"]
    Wb, bb = sd[prefix + "box_embedding.weight"], sd[prefix + "box_embedding.bias"]
    x = E[batch["categories"]] + batch["boxes"].to(E.dtype) @ Wb.t() + bb
    if "scores" in batch:  # branch on key presence, models.py:33
        Ws, bs = sd[prefix + "score_embeddings.weight"], sd[prefix + "score_embeddings.bias"]
        x = x + batch["scores"].to(E.dtype).unsqueeze(-1) * Ws[:, 0] + bs
    return layer_norm(x, sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"], eps)


def attention_core(qkv: torch.Tensor, mask_add: torch.Tensor, H: int, drop: Optional["Dropout"] = None, site: int = 0
                   ) -> torch.Tensor:
    """softmax(QK^T/sqrt(dh) + M) V per head.  qkv (S,L,3d) packed [q;k;v]; mask_add (S,L,L) of {0,-inf}."""
    S, L, d3 = qkv.shape
    d = d3 // 3
    dh = d // H
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(S, L, H, dh).transpose(1, 2)
    k = k.reshape(S, L, H, dh).transpose(1, 2)
    v = v.reshape(S, L, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh)) + mask_add.unsqueeze(1)
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)  # fully-masked rows -> 0 (torch>=2 behaviour; cannot occur under §8b invariants)
    if drop is not None:
        p = drop.attention(site, p)
    return (p @ v).transpose(1, 2).reshape(S, L, d)


def encoder_layer(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, mask_add: torch.Tensor, H: int,
                  drop: Optional["Dropout"] = None, site0: int = 0) -> torch.Tensor:
    """nn.TransformerEncoderLayer as configured at models.py:46-52,118-124:
    post-norm, gelu, dim_feedforward=4d, LN eps = torch default 1e-5 (config eps is NOT forwarded)."""
    p = lambda k: sd[prefix + k]
    qkv = x @ p("self_attn.in_proj_weight").t() + p("self_attn.in_proj_bias")
    dz = (lambda k, t: drop.elementwise(site0 + k, t)) if drop is not None else (lambda k, t: t)
    a = attention_core(qkv, mask_add, H, drop, site0)
    x = layer_norm(x + dz(1, a @ p("self_attn.out_proj.weight").t() + p("self_attn.out_proj.bias")),
                   p("norm1.weight"), p("norm1.bias"), 1e-5)
    h = dz(2, gelu(x @ p("linear1.weight").t() + p("linear1.bias")))
    x = layer_norm(x + dz(3, h @ p("linear2.weight").t() + p("linear2.bias")), p("norm2.weight"), p("norm2.bias"), 1e-5)
    return x


def _neg_inf_mask(masked: torch.Tensor, dtype) -> torch.Tensor:
    return torch.zeros(masked.shape, dtype=dtype).masked_fill(masked, float("-inf"))


def backbone_forward(sd: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], num_heads: int,
                     layer_norm_eps: float = 1e-12, prefix: str = "", dtype=torch.float32,
                     taps: Optional[Dict[str, torch.Tensor]] = None, drop: Optional["Dropout"] = None) -> torch.Tensor:
    """StltBackbone.forward — models.py:136-152 — returned batch-major (B,T,d)
    (the reference returns the (T,B,d) transpose of this)."""
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items() if k.startswith(prefix)}
    FE = prefix + "frames_embeddings."
    LE = FE + "layout_embedding."
    B, T, N = batch["categories"].shape
    # --- SpatialTransformer.forward models.py:57-81
    x = category_box_embeddings(sd, LE + "category_box_embeddings.", batch, layer_norm_eps)  # (B,T,N,d)
    if drop is not None:
        x = drop.elementwise(SITE_EMBED, x)
    if taps is not None:
        taps["embed"] = x.clone()
    d = x.shape[-1]
    x = x.reshape(B * T, N, d)
    kpm = batch["src_key_padding_mask_boxes"].reshape(B * T, N)
    m_sp = _neg_inf_mask(kpm[:, None, :].expand(B * T, N, N), x.dtype)
    n_sp = 0
    while f"{LE}transformer.layers.{n_sp}.norm1.weight" in sd:
        x = encoder_layer(x, sd, f"{LE}transformer.layers.{n_sp}.", m_sp, num_heads, drop, 8 * (n_sp + 1))
        if taps is not None:
            taps[f"spatial{n_sp}"] = x.reshape(B, T, N, d).clone()
        n_sp += 1
    f = x.reshape(B, T, N, d)[:, :, 0, :]  # CLS token per frame, models.py:79
    # --- FramesEmbeddings.forward models.py:98-111
    P = sd[FE + "position_embeddings.weight"]
    F_ = sd[FE + "frame_type_embedding.weight"]
    g = layer_norm(f + P[:T][None] + F_[batch["frame_types"]], sd[FE + "layer_norm.weight"],
                   sd[FE + "layer_norm.bias"], layer_norm_eps)
    if drop is not None:
        g = drop.elementwise(SITE_FRAMES, g)
    if taps is not None:
        taps["frames"] = g.clone()
    # --- temporal transformer models.py:140-150; causal mask utils/model_utils.py:4-7 (True strictly above diag)
    causal = torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1)
    masked = causal[None] | batch["src_key_padding_mask_frames"][:, None, :]
    m_tp = _neg_inf_mask(masked, g.dtype)
    n_tp = 0
    x = g
    while f"{prefix}transformer.layers.{n_tp}.norm1.weight" in sd:
        x = encoder_layer(x, sd, f"{prefix}transformer.layers.{n_tp}.", m_tp, num_heads, drop, 8 * (n_sp + n_tp + 1))
        if taps is not None:
            taps[f"temporal{n_tp}"] = x.clone()
        n_tp += 1
    return x  # (B,T,d)


def head_forward(sd: Dict[str, torch.Tensor], h: torch.Tensor, layer_norm_eps: float = 1e-12,
                 prefix: str = "prediction_head.") -> torch.Tensor:
    """ClassificationHead.forward — models.py:162-163."""
    p = lambda k: sd[prefix + k].to(h.dtype)
    z = gelu(h @ p("fc1.weight").t() + p("fc1.bias"))
    z = layer_norm(z, p("layer_norm.weight"), p("layer_norm.bias"), layer_norm_eps)
    return z @ p("fc2.weight").t() + p("fc2.bias")


def stlt_forward(sd: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], num_heads: int,
                 layer_norm_eps: float = 1e-12, dtype=torch.float32,
                 taps: Optional[Dict[str, torch.Tensor]] = None, drop: Optional["Dropout"] = None) -> Dict[str, torch.Tensor]:
    """Stlt.forward — models.py:185-195. -> {"stlt": (B,num_classes)}"""
    out = backbone_forward(sd, batch, num_heads, layer_norm_eps, prefix="backbone.", dtype=dtype, taps=taps, drop=drop)
    B = out.shape[0]
    h = out[torch.arange(B), batch["lengths"] - 1]  # models.py:189-192
    logits = head_forward(sd, h, layer_norm_eps)
    return {"stlt": logits}
