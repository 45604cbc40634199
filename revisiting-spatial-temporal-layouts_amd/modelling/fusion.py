"""Drop-in ``CrossAttentionFusion`` (CAF) and ``CrossAttentionCentralNetFusion`` (CACNF) for inference on PRECOMPUTED
appearance features (reference src/modelling/models.py:230-271, 286-298, 328-549; BASELINE config 5).

Differences from the reference, by design of the scope:
  * the R3D-50 trunk does not run: the batch carries ``appearance_features`` (B, 2048, 2, 4, 4) — what
    ``Resnet3D.forward_features`` returns — instead of ``video_frames``; the state dict therefore has every reference
    key EXCEPT ``…appearance_branch.resnet.*`` (load reference checkpoints with ``strict=False``);
  * inference only (no autograd path for the fusion layers yet).
As for STLT, the modules only hold parameters; the arithmetic is ``stlt_caf_forward`` in libstlt_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from .configs import MultimodalModelConfig
from .models import ClassificationHead, StltBackbone, _dev_ptr, _EncoderLayerParams, _EncoderStack, _prep_inputs, _SelfAttnParams, _Workspace


class _AttnBlock(nn.Module):
    """SelfAttentionLayer / CrossAttentionLayer parameters: ``attn`` (MultiheadAttention) + ``ln`` (models.py:345-382)."""

    def __init__(self, config):
        super().__init__()
        self.attn = _SelfAttnParams(config.hidden_size)
        self.ln = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def c_struct(self):
        ts = (self.attn.in_proj_weight, self.attn.in_proj_bias, self.attn.out_proj.weight, self.attn.out_proj.bias,
              self.ln.weight, self.ln.bias)
        return L.AttnBlockParams(*[_dev_ptr(t) for t in ts])


class FeedforwardModule(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.linear1 = nn.Linear(config.hidden_size, config.hidden_size * 4)
        self.linear2 = nn.Linear(config.hidden_size * 4, config.hidden_size)
        self.ln = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def c_struct(self):
        ts = (self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.ln.weight, self.ln.bias)
        return L.FfnBlockParams(*[_dev_ptr(t) for t in ts])


class CrossModalModule(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.cross_attn = _AttnBlock(config)
        self.layout_attn = _AttnBlock(config)
        self.layout_ffn = FeedforwardModule(config)
        self.appearance_attn = _AttnBlock(config)
        self.appearance_ffn = _AttnBlock(config)  # a SelfAttentionLayer in the reference (models.py:401)

    def c_struct(self):
        return L.CrossModalParams(self.cross_attn.c_struct(), self.layout_attn.c_struct(), self.appearance_attn.c_struct(),
                                  self.appearance_ffn.c_struct(), self.layout_ffn.c_struct())


class TransformerResnetFeatures(nn.Module):
    """``TransformerResnet`` minus the R3D trunk (models.py:230-252): projector, CLS token, position table, ReLU encoder."""

    def __init__(self, config):
        super().__init__()
        d = config.hidden_size
        self.projector = nn.Conv3d(2048, d, kernel_size=(1, 1, 1))
        self.transformer = _EncoderStack(_EncoderLayerParams(d), config.num_appearance_layers)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, d))
        self.pos_embed = nn.Parameter(torch.zeros(config.appearance_num_frames + 1, 1, d))
        self.classifier = nn.Linear(d, config.num_classes)  # unused by CAF/CACNF, present in reference checkpoints


class FusionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size * 2, config.hidden_size)
        self.layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.fc2 = nn.Linear(config.hidden_size, config.num_classes)


def _head_struct(h) -> "L.HeadParams":
    if h is None:
        return L.HeadParams()
    return L.HeadParams(*[_dev_ptr(t) for t in (h.fc1.weight, h.fc1.bias, h.layer_norm.weight, h.layer_norm.bias,
                                                h.fc2.weight, h.fc2.bias)])


class CrossAttentionFusionBackbone(nn.Module):
    def __init__(self, config: MultimodalModelConfig):
        super().__init__()
        self.config = config
        self.layout_branch = StltBackbone(config.stlt_config)
        self.appearance_branch = TransformerResnetFeatures(config)
        self.mm_fusion = nn.ModuleList([CrossModalModule(config) for _ in range(config.num_fusion_layers)])
        self._ws = _Workspace()

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_ws"] = _Workspace()
        return state

    def run(self, batch: Dict[str, torch.Tensor], fusion_head, layout_head=None, appearance_head=None):
        """-> (logits_caf, logits_stlt | None, logits_resnet3d | None, logits_ensemble | None)"""
        if self.training and self.config.hidden_dropout_prob > 0:
            raise L.StltHipError("CAF / CACNF are inference-only in this build: call model.train(False)")
        lib = L.load()
        inp, keep, (B, T, N) = _prep_inputs(batch, need_lengths=True)
        feats = ops._chk(batch["appearance_features"].contiguous(), torch.float32, "appearance_features")
        ab = self.appearance_branch
        Cc = ab.projector.weight.shape[1]
        S = ab.pos_embed.shape[0] - 1
        if feats.shape[0] != B or feats.shape[1] != Cc or feats[0, 0].numel() != S:
            raise L.StltHipError(f"appearance_features must be (B, {Cc}, ...{S} positions), got {tuple(feats.shape)}")
        device = feats.device
        cfg = self.config
        K = fusion_head.fc2.weight.shape[0]
        lay, sp, tp = self.layout_branch._build_struct(None, _dev_ptr)
        lay.n_classes = K
        app = (L.LayerParams * max(1, len(ab.transformer.layers)))(*[l.c_struct() for l in ab.transformer.layers])
        fus = (L.CrossModalParams * max(1, len(self.mm_fusion)))(*[m.c_struct() for m in self.mm_fusion])
        p = L.CafParams()
        p.layout = lay
        p.feat_channels, p.app_tokens = Cc, S
        p.proj_w, p.proj_b = _dev_ptr(ab.projector.weight), _dev_ptr(ab.projector.bias)
        p.cls_token, p.pos_embed = _dev_ptr(ab.cls_token), _dev_ptr(ab.pos_embed)
        p.n_app_layers, p.app_layers = len(ab.transformer.layers), app
        p.n_fusion, p.fusion = len(self.mm_fusion), fus
        p.fusion_head, p.layout_head, p.appearance_head = _head_struct(fusion_head), _head_struct(layout_head), _head_struct(appearance_head)
        nbytes = int(lib.stlt_caf_workspace_bytes(B, T, N, cfg.hidden_size, Cc, S, K))
        ws = self._ws.get(nbytes, device)
        outs = [torch.empty(B, K, device=device, dtype=torch.float32) for _ in range(4 if layout_head is not None else 1)]
        ptrs = [o.data_ptr() for o in outs] + [None] * (4 - len(outs))
        with torch.cuda.device(device):
            L.check(lib.stlt_caf_forward(C.byref(p), C.byref(inp), feats.data_ptr(), ws.data_ptr(), ws.numel(), ptrs[0], ptrs[1],
                                         ptrs[2], ptrs[3], torch.cuda.current_stream().cuda_stream), "stlt_caf_forward")
        return outs


class CrossAttentionFusion(nn.Module):
    """CAF (models.py:486-498): ``forward(batch) -> {"caf": (B, num_classes)}``."""

    def __init__(self, config: MultimodalModelConfig):
        super().__init__()
        self.caf_backbone = CrossAttentionFusionBackbone(config)
        self.classifier = FusionHead(config)
        self.logit_names = ("caf",)

    @torch.no_grad()
    def forward(self, batch: Dict[str, torch.Tensor]):
        (caf,) = self.caf_backbone.run(batch, self.classifier)
        return {"caf": caf}


class CrossAttentionCentralNetFusion(nn.Module):
    """CACNF (models.py:501-549): ``forward(batch) -> {"stlt", "resnet3d", "caf", "ensemble"}``."""

    def __init__(self, config: MultimodalModelConfig):
        super().__init__()
        self.config = config
        self.backbone = CrossAttentionFusionBackbone(config)
        self.layout_classifier = ClassificationHead(config)
        self.appearance_classifier = ClassificationHead(config)
        self.fusion_classifier = FusionHead(config)
        self.logit_names = ("stlt", "resnet3d", "caf", "ensemble")

    @torch.no_grad()
    def forward(self, batch: Dict[str, torch.Tensor]):
        caf, stlt, res, ens = self.backbone.run(batch, self.fusion_classifier, self.layout_classifier, self.appearance_classifier)
        return {"stlt": stlt, "resnet3d": res, "caf": caf, "ensemble": ens}


from .models import models_factory  # noqa: E402

models_factory["caf"] = CrossAttentionFusion
models_factory["cacnf"] = CrossAttentionCentralNetFusion
