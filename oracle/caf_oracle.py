"""CPU oracle of CAF / CACNF on precomputed appearance features.  TEST INFRASTRUCTURE ONLY (see stlt_oracle.py header).

Plain-tensor restatement of the reference's CrossAttentionFusionBackbone / CrossAttentionFusion /
CrossAttentionCentralNetFusion (src/modelling/models.py:434-549) with the appearance branch starting from the feature
map that Resnet3D.forward_features returns (models.py:221-222, 253-271).  Pinned by tests/golden/caf_*.npz, captured from
the reference's own modules (tools/gen_golden_caf.py).
"""
import math
from typing import Dict

import torch

from . import stlt_oracle as O


def mha(sd, pre, q_in, kv_in, H, kpm_k=None, causal=False):
    """nn.MultiheadAttention forward (batch-major here): q_in (B,Lq,d), kv_in (B,Lk,d) -> (B,Lq,d)."""
    W, b = sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"]
    d = q_in.shape[-1]
    q = q_in @ W[:d].t() + b[:d]
    k = kv_in @ W[d:2 * d].t() + b[d:2 * d]
    v = kv_in @ W[2 * d:].t() + b[2 * d:]
    B, Lq, Lk, dh = q.shape[0], q.shape[1], k.shape[1], d // H
    qh = q.reshape(B, Lq, H, dh).transpose(1, 2)
    kh = k.reshape(B, Lk, H, dh).transpose(1, 2)
    vh = v.reshape(B, Lk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if kpm_k is not None:
        s = s.masked_fill(kpm_k[:, None, None, :], float("-inf"))
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Lq, Lk, dtype=torch.bool), diagonal=1), float("-inf"))
    o = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, d)
    return o @ sd[pre + "out_proj.weight"].t() + sd[pre + "out_proj.bias"]


def attn_layer(sd, pre, x, ctx, H, eps, kpm_k=None, causal=False):
    """SelfAttentionLayer (ctx is x) / CrossAttentionLayer, models.py:345-382."""
    return O.layer_norm(mha(sd, pre + "attn.", x, ctx, H, kpm_k, causal) + x, sd[pre + "ln.weight"], sd[pre + "ln.bias"], eps)


def appearance_forward(sd, pre, feats, H):
    """TransformerResnet.forward_features from the feature map on, models.py:257-271. -> (B, S+1, d)"""
    B, Cc = feats.shape[0], feats.shape[1]
    Wp = sd[pre + "projector.weight"].reshape(-1, Cc)
    x = feats.flatten(2).transpose(1, 2) @ Wp.t() + sd[pre + "projector.bias"]            # (B,S,d)
    x = torch.cat((sd[pre + "cls_token"].reshape(1, 1, -1).expand(B, -1, -1), x), dim=1)
    x = x + sd[pre + "pos_embed"].reshape(1, -1, x.shape[-1])
    l = 0
    while f"{pre}transformer.layers.{l}.norm1.weight" in sd:  # ReLU encoder layers (nn.TransformerEncoderLayer default)
        p = lambda k: sd[f"{pre}transformer.layers.{l}.{k}"]
        a = mha({k[len(f"{pre}transformer.layers.{l}.self_attn."):]: v for k, v in sd.items()
                 if k.startswith(f"{pre}transformer.layers.{l}.self_attn.")}, "", x, x, H)
        x = O.layer_norm(x + a, p("norm1.weight"), p("norm1.bias"), 1e-5)
        h = torch.relu(x @ p("linear1.weight").t() + p("linear1.bias"))
        x = O.layer_norm(x + h @ p("linear2.weight").t() + p("linear2.bias"), p("norm2.weight"), p("norm2.bias"), 1e-5)
        l += 1
    return x


def backbone(sd, pre, batch, H, eps):
    """CrossAttentionFusionBackbone.forward, models.py:446-483 (batch-major)."""
    Lh = O.backbone_forward(sd, batch, H, eps, prefix=pre + "layout_branch.")
    Ah = appearance_forward(sd, pre + "appearance_branch.", batch["appearance_features"], H)
    B = Lh.shape[0]
    idx = torch.arange(B)
    lay_state, app_state = Lh[idx, batch["lengths"] - 1], Ah[:, 0]
    kpm = batch["src_key_padding_mask_frames"]
    l = 0
    while f"{pre}mm_fusion.{l}.cross_attn.ln.weight" in sd:  # CrossModalModule.forward, models.py:403-431
        m = f"{pre}mm_fusion.{l}."
        la = attn_layer(sd, m + "cross_attn.", Lh, Ah, H, eps)
        aa = attn_layer(sd, m + "cross_attn.", Ah, Lh, H, eps, kpm_k=kpm)
        la = attn_layer(sd, m + "layout_attn.", la, la, H, eps, kpm_k=kpm, causal=True)
        aa = attn_layer(sd, m + "appearance_attn.", aa, aa, H, eps)
        f = O.gelu(la @ sd[m + "layout_ffn.linear1.weight"].t() + sd[m + "layout_ffn.linear1.bias"])
        f = f @ sd[m + "layout_ffn.linear2.weight"].t() + sd[m + "layout_ffn.linear2.bias"]
        Lh = O.layer_norm(f + la, sd[m + "layout_ffn.ln.weight"], sd[m + "layout_ffn.ln.bias"], eps)
        Ah = attn_layer(sd, m + "appearance_ffn.", aa, aa, H, eps)  # appearance_ffn is a SelfAttentionLayer (models.py:401)
        l += 1
    fused = torch.cat((Lh[idx, batch["lengths"] - 1], Ah[:, 0]), dim=-1)
    return lay_state, app_state, fused


def caf_forward(sd, batch, H, eps=1e-12) -> Dict[str, torch.Tensor]:
    _, _, fused = backbone(sd, "caf_backbone.", batch, H, eps)
    return {"caf": O.head_forward(sd, fused, eps, prefix="classifier.")}


def cacnf_forward(sd, batch, H, eps=1e-12) -> Dict[str, torch.Tensor]:
    lay, app, fused = backbone(sd, "backbone.", batch, H, eps)
    out = {"stlt": O.head_forward(sd, lay, eps, prefix="layout_classifier."),
           "resnet3d": O.head_forward(sd, app, eps, prefix="appearance_classifier."),
           "caf": O.head_forward(sd, fused, eps, prefix="fusion_classifier.")}
    out["ensemble"] = (out["stlt"] + out["resnet3d"] + out["caf"]) / 3
    return out


def lcf_forward(sd, batch, H, eps=1e-12) -> Dict[str, torch.Tensor]:
    """LateConcatenationFusion.forward, models.py:296-322: FusionHead on [layout state at lengths-1 ; appearance CLS state]
    (the backbone above with no cross-modal module and no key prefix)."""
    _, _, fused = backbone(sd, "", batch, H, eps)
    return {"lcf": O.head_forward(sd, fused, eps, prefix="classifier.")}
