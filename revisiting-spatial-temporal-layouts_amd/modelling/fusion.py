"""Drop-in ``CrossAttentionFusion`` (CAF), ``CrossAttentionCentralNetFusion`` (CACNF) and ``LateConcatenationFusion`` (LCF)
on PRECOMPUTED appearance features (reference src/modelling/models.py:230-271, 286-322, 328-549; BASELINE config 5).

Differences from the reference, by design of the scope:
  * the R3D-50 trunk does not run: the batch carries ``appearance_features`` (B, 2048, 2, 4, 4) — what
    ``Resnet3D.forward_features`` returns — instead of ``video_frames``; the state dict therefore has every reference
    key EXCEPT ``…appearance_branch.resnet.*`` (load reference checkpoints with ``strict=False``);
  * inference is one native call (``stlt_caf_forward``).  Training — with autograd enabled and trainable parameters —
    composes the same arithmetic from the op-level autograd Functions of ``ops.py`` (native forward AND backward kernels
    per op: linear, attention, add+LayerNorm, GELU, the two embedding kernels); a frozen layout branch runs through the
    native forward without a tape, a trainable one through ``StltBackbone.forward_train``.  Dropout (reference
    models.py:333,341,350,358,368,376 and the appearance encoder's fixed 0.1) is the native counter-based mask everywhere
    (``ops.dropout`` at the post-attention / feed-forward sites, inside the attention kernel on the probabilities); the
    appearance encoder's ReLU runs in its product's epilogue.
As for STLT, the modules only hold parameters.
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

    # ---- training path: op-level autograd over native kernels -------------------------------------------------
    @staticmethod
    def _lin(x, lin):
        return ops.LinearFn.apply(x, lin.weight, lin.bias)

    def _attn_block(self, blk: _AttnBlock, x, ctx, kpm, causal, p_drop):
        """SelfAttentionLayer (ctx is x) / CrossAttentionLayer, models.py:345-382: LN(dropout(MHA(x, ctx, ctx)) + x) — one native
        call each way (ops.AttnBlockFn)."""
        return ops.AttnBlockFn.apply(x, None if ctx is x else ctx, kpm, causal, self.config.num_attention_heads, self.config.layer_norm_eps,
                                     p_drop if self.training else 0.0, blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.attn.out_proj.weight,
                                     blk.attn.out_proj.bias, blk.ln.weight, blk.ln.bias)

    def _appearance_train(self, feats):
        """TransformerResnet.forward_features from the feature map on (models.py:257-271), batch-major (B, S+1, d)."""
        ab = self.appearance_branch
        B, Cc = feats.shape[0], feats.shape[1]
        d = ab.projector.weight.shape[0]
        x = ops.LinearFn.apply(feats.flatten(2).transpose(1, 2).contiguous(), ab.projector.weight.view(d, Cc), ab.projector.bias)
        x = torch.cat((ab.cls_token.view(1, 1, d).expand(B, -1, -1), x), dim=1) + ab.pos_embed.view(1, -1, d)
        H = self.config.num_attention_heads
        p = 0.1 if self.training else 0.0
        for l in ab.transformer.layers:  # nn.TransformerEncoderLayer defaults: ReLU, post-norm, eps 1e-5, dropout 0.1
            sa = l.self_attn
            x = ops.AttnBlockFn.apply(x, None, None, False, H, 1e-5, p, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                                      l.norm1.weight, l.norm1.bias)
            x = ops.FfnBlockFn.apply(x, 1e-5, L.ACT_RELU, True, p, l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias,
                                     l.norm2.weight, l.norm2.bias)
        return x

    def _head_train(self, h, x):
        eps = self.config.layer_norm_eps
        z = ops.GeluFn.apply(self._lin(x.contiguous(), h.fc1))
        z = ops.AddLayerNormFn.apply(z, None, h.layer_norm.weight, h.layer_norm.bias, eps)
        return self._lin(z, h.fc2)

    def run_train(self, batch: Dict[str, torch.Tensor], fusion_head, layout_head=None, appearance_head=None):
        """Differentiable forward (see the module docstring).  -> same tuple as run()."""
        feats = ops._chk(batch["appearance_features"].contiguous(), torch.float32, "appearance_features")
        if any(q.requires_grad for q in self.layout_branch.parameters()):
            Lh = self.layout_branch.forward_train(batch)  # (B,T,d) with its autograd graph (op-level composition)
        else:  # frozen (load_backbone_path + freeze_backbone): one native call, no tape
            was_training = self.layout_branch.training
            self.layout_branch.train(False)
            with torch.no_grad():
                Lh = self.layout_branch.forward_batch_major(batch)
            self.layout_branch.train(was_training)
        Ah = self._appearance_train(feats)
        B, d = Lh.shape[0], Lh.shape[2]
        # row lengths - 1 of every clip (models.py:455-459) as a gather: its backward is one scatter-add, where advanced indexing
        # (Lh[arange(B), last]) sorts the indices and runs six launches.  Only a NEGATIVE row wraps around, as indexing wraps it; a row beyond
        # the last frame stays out of range and the gather's bounds check fails loudly, as the reference's indexing raises IndexError
        last = batch["lengths"].to(Lh.device) - 1
        last = torch.where(last < 0, last + Lh.shape[1], last).view(B, 1, 1).expand(B, 1, d)
        lay_state, app_state = Lh.gather(1, last).squeeze(1), Ah[:, 0]
        kpm = batch["src_key_padding_mask_frames"]
        p = self.config.hidden_dropout_prob
        eps = self.config.layer_norm_eps
        for m in self.mm_fusion:  # CrossModalModule.forward, models.py:403-431
            la = self._attn_block(m.cross_attn, Lh, Ah, None, False, p)
            aa = self._attn_block(m.cross_attn, Ah, Lh, kpm, False, p)
            la = self._attn_block(m.layout_attn, la, la, kpm, True, p)
            aa = self._attn_block(m.appearance_attn, aa, aa, None, False, p)
            ff = m.layout_ffn
            Lh = ops.FfnBlockFn.apply(la, eps, L.ACT_GELU, False, p if self.training else 0.0, ff.linear1.weight, ff.linear1.bias, ff.linear2.weight,
                                      ff.linear2.bias, ff.ln.weight, ff.ln.bias)
            Ah = self._attn_block(m.appearance_ffn, aa, aa, None, False, p)
        fused = torch.cat((Lh.gather(1, last).squeeze(1), Ah[:, 0]), dim=-1)
        caf = self._head_train(fusion_head, fused)
        if layout_head is None:
            return (caf,)
        stlt = self._head_train(layout_head, lay_state)
        res = self._head_train(appearance_head, app_state)
        return caf, stlt, res, (stlt + res + caf) / 3

    def run(self, batch: Dict[str, torch.Tensor], fusion_head, layout_head=None, appearance_head=None):
        """-> (logits_caf, logits_stlt | None, logits_resnet3d | None, logits_ensemble | None)"""
        heads = [h for h in (fusion_head, layout_head, appearance_head) if h is not None]
        if torch.is_grad_enabled() and any(q.requires_grad for mod in [self] + heads for q in mod.parameters()):
            return self.run_train(batch, fusion_head, layout_head, appearance_head)
        if self.training and self.config.hidden_dropout_prob > 0:
            # train mode without grad: nn.Dropout is still live in the reference (models.py:334,351,376): the training
            # composition runs (its Functions just execute their forward kernels under no_grad)
            return self.run_train(batch, fusion_head, layout_head, appearance_head)
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
            # layout_branch.skip_padding (as on a stand-alone StltBackbone): the layout branch on the real tokens / frames only
            flags = L.FLAG_SKIP_PADDING if self.layout_branch.skip_padding else 0
            L.check(lib.stlt_caf_forward_flags(C.byref(p), C.byref(inp), feats.data_ptr(), ws.data_ptr(), ws.numel(), flags, ptrs[0],
                                               ptrs[1], ptrs[2], ptrs[3], torch.cuda.current_stream().cuda_stream), "stlt_caf_forward")
        return outs


class CrossAttentionFusion(nn.Module):
    """CAF (models.py:486-498): ``forward(batch) -> {"caf": (B, num_classes)}``."""

    def __init__(self, config: MultimodalModelConfig):
        super().__init__()
        self.caf_backbone = CrossAttentionFusionBackbone(config)
        self.classifier = FusionHead(config)
        self.logit_names = ("caf",)

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

    def forward(self, batch: Dict[str, torch.Tensor]):
        caf, stlt, res, ens = self.backbone.run(batch, self.fusion_classifier, self.layout_classifier, self.appearance_classifier)
        return {"stlt": stlt, "resnet3d": res, "caf": caf, "ensemble": ens}


class LateConcatenationFusion(nn.Module):
    """LCF (reference models.py:296-322): ``forward(batch) -> {"lcf": (B, num_classes)}`` — the FusionHead on the layout
    state at ``lengths-1`` concatenated with the appearance branch's CLS state.  That is the fusion backbone with zero
    cross-modal layers, so it runs through the same native entry point (``stlt_caf_forward``, ``n_fusion = 0``) and the same
    op-level training composition.  State-dict keys are the reference's (``layout_branch.*``, ``appearance_branch.*``,
    ``classifier.*``) minus ``…resnet.*``: the helper that owns the launch logic shares these modules without being
    registered as a child."""

    def __init__(self, config: MultimodalModelConfig):
        super().__init__()
        self.config = config
        self.layout_branch = StltBackbone(config.stlt_config)
        self.appearance_branch = TransformerResnetFeatures(config)
        self.classifier = FusionHead(config)
        self.logit_names = ("lcf",)
        object.__setattr__(self, "_runner", self._make_runner())

    def _make_runner(self):
        r = CrossAttentionFusionBackbone.__new__(CrossAttentionFusionBackbone)
        nn.Module.__init__(r)
        r.config = self.config
        r.layout_branch, r.appearance_branch = self.layout_branch, self.appearance_branch  # shared, not copied
        r.mm_fusion = nn.ModuleList([])
        r._ws = _Workspace()
        return r

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_runner", None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        object.__setattr__(self, "_runner", self._make_runner())

    def train(self, mode: bool = True):
        super().train(mode)
        self._runner.training = mode  # the runner is not a child: keep its mode flag (dropout switch) in step
        return self

    def forward(self, batch: Dict[str, torch.Tensor]):
        (lcf,) = self._runner.run(batch, self.classifier)
        return {"lcf": lcf}


from .models import models_factory  # noqa: E402

models_factory["caf"] = CrossAttentionFusion
models_factory["cacnf"] = CrossAttentionCentralNetFusion
models_factory["lcf"] = LateConcatenationFusion
