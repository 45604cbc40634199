"""CAF / CACNF / LCF on precomputed appearance features (BASELINE config 5) against fixtures captured
from the reference's own modules (tools/gen_golden_caf.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import caf_oracle as CO

NAME = "cfg1"
EXTRA = dict(appearance_num_frames=32, num_appearance_layers=2, num_fusion_layers=2)


def _case(synth, model_name):
    z = np.load(os.path.join(GOLDEN, f"{model_name}_cfg1.npz"))
    meta = json.load(open(os.path.join(GOLDEN, f"{model_name}_cfg1_schema.json")))
    c = synth.CONFIGS[NAME]
    sd = synth.make_state_dict({k: tuple(v) for k, v in meta["keys"].items()}, seed=meta["weight_seed"])
    batch = synth.make_batch(meta["batch"], c["T"], c["N"], seed=meta["input_seed"])
    batch["appearance_features"] = synth.make_appearance_features(meta["batch"], seed=meta["feature_seed"])
    return z, meta, sd, batch, c


@pytest.mark.parametrize("model_name", ["caf", "cacnf", "lcf"])
def test_caf_oracle_matches_reference(synth, model_name):
    z, meta, sd, batch, c = _case(synth, model_name)
    fwd = {"caf": CO.caf_forward, "cacnf": CO.cacnf_forward, "lcf": CO.lcf_forward}[model_name]
    with torch.no_grad():
        out = fwd(sd, batch, c["num_attention_heads"])
    assert set(out) == set(z.files)
    for k in z.files:
        assert np.abs(out[k].numpy() - z[k]).max() <= 3e-5, k


@pytest.mark.parametrize("model_name", ["caf", "cacnf", "lcf"])
def test_caf_state_dict_keys_are_the_reference_non_r3d_keys(pkg, model_name):
    _, meta, _, _, _ = _case(pkg.synth, model_name)
    cls = pkg.models_factory[model_name]
    m = cls(pkg.MultimodalModelConfig(**dict(pkg.synth.model_kwargs(NAME), **EXTRA)))
    sd = m.state_dict()
    assert list(sd) == list(meta["keys"])
    assert all(list(sd[k].shape) == meta["keys"][k] for k in sd)


def _set_skip_padding(pkg, m, on):
    """The fusion models take the flag from their layout branch (a StltBackbone), like a stand-alone backbone."""
    n = 0
    for mod in m.modules():
        if isinstance(mod, pkg.StltBackbone):
            mod.skip_padding = on
            n += 1
    assert n == 1


@pytest.mark.gpu
@pytest.mark.parametrize("skip_padding", [False, True])
@pytest.mark.parametrize("model_name", ["caf", "cacnf", "lcf"])
def test_caf_gpu_matches_reference(pkg, model_name, skip_padding):
    z, meta, sd, batch, c = _case(pkg.synth, model_name)
    cls = pkg.models_factory[model_name]
    m = cls(pkg.MultimodalModelConfig(**dict(pkg.synth.model_kwargs(NAME), **EXTRA)))
    m.load_state_dict(sd, strict=True)
    m.train(False).to("cuda")
    _set_skip_padding(pkg, m, skip_padding)  # layout branch on the real tokens / frames only: same logits
    with torch.no_grad():
        out = m({k: v.to("cuda") for k, v in batch.items()})
    assert tuple(out) == m.logit_names or set(out) == set(m.logit_names)
    for k in z.files:
        got = out[k].cpu().numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - z[k]).max() <= 1e-4, k


@pytest.mark.gpu
def test_caf_gpu_full_width_matches_oracle(pkg):
    """d=768 / 12 heads / T=32, N=7 (the cfg2 layout shapes of BASELINE config 5), default 4+4 fusion/appearance layers."""
    c = pkg.synth.CONFIGS["cfg2"]
    kw = dict(pkg.synth.model_kwargs("cfg2"), appearance_num_frames=32)
    m = pkg.CrossAttentionCentralNetFusion(pkg.MultimodalModelConfig(**kw))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    m.load_state_dict(sd)
    m.train(False).to("cuda")
    batch = pkg.synth.make_batch(2, c["T"], c["N"], seed=8)
    batch["appearance_features"] = pkg.synth.make_appearance_features(2, seed=9)
    with torch.no_grad():
        out = m({k: v.to("cuda") for k, v in batch.items()})
        ref = CO.cacnf_forward(sd, batch, c["num_attention_heads"])
        _set_skip_padding(pkg, m, True)
        out_sp = m({k: v.to("cuda") for k, v in batch.items()})
    for k in ref:
        assert (out[k].cpu() - ref[k]).abs().max().item() <= 1e-4, k
        assert (out_sp[k].cpu() - ref[k]).abs().max().item() <= 1e-4, k
        assert (out_sp[k] - out[k]).abs().max().item() <= 2e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("freeze_layout", [True, False])
@pytest.mark.parametrize("model_name", ["caf", "cacnf"])
def test_fusion_training_path_matches_golden_and_oracle_gradients(pkg, model_name, freeze_layout):
    """With autograd on, the forward is composed from the op-level autograd Functions (native forward and backward
    kernels); a frozen layout branch runs natively without a tape, a trainable one through StltBackbone.forward_train.
    The logits must still be the reference's, and every trainable gradient must match torch autograd on the CPU oracle."""
    import torch.nn.functional as F
    z, meta, sd, batch, c = _case(pkg.synth, model_name)
    cls = pkg.models_factory[model_name]
    m = cls(pkg.MultimodalModelConfig(**dict(pkg.synth.model_kwargs(NAME), **EXTRA)))
    m.load_state_dict(sd, strict=True)
    m.train(False).to("cuda")  # eval mode: no dropout, the appearance encoder's fixed 0.1 included
    bb = m.caf_backbone if model_name == "caf" else m.backbone
    dev_batch = {k: v.to("cuda") for k, v in batch.items()}
    if freeze_layout:
        for q in bb.layout_branch.parameters():
            q.requires_grad_(False)
    out = m(dev_batch)
    for k in z.files:
        assert out[k].requires_grad and np.abs(out[k].detach().cpu().numpy() - z[k]).max() <= 1e-4, k
    labels = torch.randint(0, 174, (meta["batch"],), generator=torch.Generator().manual_seed(1))
    loss = sum(F.cross_entropy(v, labels.to("cuda")) for v in out.values()) / len(out)
    loss.backward()
    frozen_prefix = ("caf_backbone." if model_name == "caf" else "backbone.") + "layout_branch."
    frozen = lambda k: freeze_layout and k.startswith(frozen_prefix)
    leaves = {k: (v.clone().requires_grad_(not frozen(k)) if v.is_floating_point() else v) for k, v in sd.items()}
    fwd = CO.caf_forward if model_name == "caf" else CO.cacnf_forward
    ref = fwd(leaves, batch, c["num_attention_heads"])  # fp32 CPU autograd (the oracle's layout branch is fp32)
    ref_loss = sum(F.cross_entropy(v, labels) for v in ref.values()) / len(ref)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5
    checked = 0
    for k, prm in m.named_parameters():
        g_ref = leaves[k].grad if leaves[k].is_floating_point() else None
        if frozen(k):
            assert prm.grad is None, k
            continue
        if prm.grad is None or g_ref is None:  # parameters the forward never reads (dead encoder_layer copy, unused classifier, scores)
            assert g_ref is None or g_ref.abs().max().item() == 0.0, k
            assert prm.grad is None or prm.grad.abs().max().item() == 0.0, k
            continue
        scale = max(g_ref.abs().max().item(), 1e-6)
        assert (prm.grad.cpu() - g_ref).abs().max().item() / scale <= 5e-4, k
        checked += 1
    assert checked > (40 if freeze_layout else 150)


@pytest.mark.gpu
def test_backbone_forward_train_matches_native_forward(pkg):
    """StltBackbone under autograd (op-level composition) against its own native no-grad forward, cfg1 with scores."""
    c = pkg.synth.CONFIGS[NAME]
    m = pkg.StltBackbone(pkg.StltModelConfig(**pkg.synth.model_kwargs(NAME)))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=9, gain=1.5))
    m.train(False).to("cuda")
    batch = {k: v.to("cuda") for k, v in pkg.synth.make_batch(3, c["T"], c["N"], seed=4, with_scores=True).items()}
    with torch.no_grad():
        ref = m(batch)
    out = m(batch)  # (T,B,d) like the reference
    assert out.requires_grad and out.shape == ref.shape
    real = (~batch["src_key_padding_mask_frames"]).t()  # padded frames' rows differ in nothing that matters, but compare the real ones
    assert (out.detach() - ref)[real].abs().max().item() <= 2e-5
    out.sum().backward()
    assert m.frames_embeddings.layout_embedding.category_box_embeddings.score_embeddings.weight.grad is not None

@pytest.mark.gpu
def test_fusion_training_step_with_dropout_runs(pkg):
    kw = dict(pkg.synth.model_kwargs(NAME), **EXTRA)
    kw["hidden_dropout_prob"] = 0.1
    m = pkg.CrossAttentionFusion(pkg.MultimodalModelConfig(**kw))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5))
    m.train(True).to("cuda")
    for q in m.caf_backbone.layout_branch.parameters():
        q.requires_grad_(False)
    c = pkg.synth.CONFIGS[NAME]
    batch = pkg.synth.make_batch(4, c["T"], c["N"], seed=8)
    batch["appearance_features"] = pkg.synth.make_appearance_features(4, seed=9)
    batch = {k: v.to("cuda") for k, v in batch.items()}
    labels = torch.tensor([1, 2, 3, 4], device="cuda")
    opt = torch.optim.AdamW([q for q in m.parameters() if q.requires_grad], lr=1e-4)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(batch)["caf"], labels)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))


@pytest.mark.gpu
def test_lcf_training_path_matches_golden_and_oracle_gradients(pkg):
    """LCF under autograd: the fusion backbone's op-level composition with zero cross-modal layers; logits = the reference's,
    gradients = torch autograd on the oracle."""
    import torch.nn.functional as F
    z, meta, sd, batch, c = _case(pkg.synth, "lcf")
    m = pkg.LateConcatenationFusion(pkg.MultimodalModelConfig(**dict(pkg.synth.model_kwargs(NAME), **EXTRA)))
    m.load_state_dict(sd, strict=True)
    m.train(False).to("cuda")
    out = m({k: v.to("cuda") for k, v in batch.items()})["lcf"]
    assert out.requires_grad and np.abs(out.detach().cpu().numpy() - z["lcf"]).max() <= 1e-4
    labels = torch.randint(0, 174, (meta["batch"],), generator=torch.Generator().manual_seed(2))
    F.cross_entropy(out, labels.to("cuda")).backward()
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    F.cross_entropy(CO.lcf_forward(leaves, batch, c["num_attention_heads"])["lcf"], labels).backward()
    checked = 0
    for k, prm in m.named_parameters():
        g_ref = leaves[k].grad
        if prm.grad is None or g_ref is None:
            assert (g_ref is None or g_ref.abs().max().item() == 0.0) and (prm.grad is None or prm.grad.abs().max().item() == 0.0), k
            continue
        assert (prm.grad.cpu() - g_ref).abs().max().item() / max(g_ref.abs().max().item(), 1e-6) <= 5e-4, k
        checked += 1
    assert checked > 150


@pytest.mark.gpu
def test_fusion_training_with_long_layouts_runs_through_the_streamed_attention_backward(pkg):
    """T = 100 frames (> 64): the layout branch's temporal attention and the cross-attention onto the layout tokens train
    through the streamed backward; gradients against torch autograd on the oracle."""
    import torch.nn.functional as F
    kw = dict(pkg.synth.model_kwargs(NAME), **EXTRA)
    kw["layout_num_frames"] = 128
    T = 100
    m = pkg.CrossAttentionFusion(pkg.MultimodalModelConfig(**kw))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=11)
    m.load_state_dict(sd)
    m.train(False).to("cuda")
    c = pkg.synth.CONFIGS[NAME]
    batch = pkg.synth.make_batch(2, T, c["N"], seed=3)
    batch["appearance_features"] = pkg.synth.make_appearance_features(2, seed=4)
    labels = torch.tensor([5, 9])
    F.cross_entropy(m({k: v.to("cuda") for k, v in batch.items()})["caf"], labels.to("cuda")).backward()
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    F.cross_entropy(CO.caf_forward(leaves, batch, c["num_attention_heads"])["caf"], labels).backward()
    checked = 0
    for k, prm in m.named_parameters():
        g_ref = leaves[k].grad
        if prm.grad is None or g_ref is None or g_ref.abs().max().item() == 0.0:
            continue
        assert (prm.grad.cpu() - g_ref).abs().max().item() / g_ref.abs().max().item() <= 1e-3, k
        checked += 1
    assert checked > 150


@pytest.mark.gpu
def test_trainer_on_a_fusion_model_matches_the_stock_loop(pkg):
    """train.Trainer on CACNF (gradients bound to one flat buffer, native block backwards accumulating into it in place, native
    criterion, fused clip + AdamW) against the reference's loop written with stock torch ops (zero_grad, F.cross_entropy per
    head, clip_grad_norm_, torch.optim.AdamW) on a twin model: same losses, same gradient norm, same parameters after 3 steps."""
    kw = dict(pkg.synth.model_kwargs(NAME), **EXTRA)
    kw["hidden_dropout_prob"] = 0.0
    c = pkg.synth.CONFIGS[NAME]
    batch = pkg.synth.make_batch(6, c["T"], c["N"], seed=21)
    batch["appearance_features"] = pkg.synth.make_appearance_features(6, seed=22)
    batch = {k: v.to("cuda") for k, v in batch.items()}
    labels = torch.randint(0, c["num_classes"], (6,), generator=torch.Generator().manual_seed(5)).cuda()
    batch["labels"] = labels
    twins = []
    for _ in range(2):
        m = pkg.CrossAttentionCentralNetFusion(pkg.MultimodalModelConfig(**kw))
        m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=7))
        m.train(False).to("cuda")  # eval-mode arithmetic (the appearance encoder's fixed 0.1 dropout included: the two loops would
        twins.append(m)            # draw different masks); gradients still flow, grad is enabled
    ours, ref = twins
    tr = pkg.train.Trainer(ours, "something", learning_rate=1e-3, weight_decay=1e-3, clip_val=5.0, warmup_steps=0, total_steps=1000)
    tr_train = ours.train
    ours.train = lambda mode=True: ours  # Trainer.step() switches to train mode: keep this twin in eval mode
    opt = torch.optim.AdamW(pkg.train.add_weight_decay(ref, 1e-3), lr=1e-3)
    sched = pkg.train.linear_schedule_with_warmup(opt, 0, 1000)
    for step in range(3):
        res = tr.step(batch)
        opt.zero_grad(set_to_none=True)
        out = ref(batch)
        loss = sum(torch.nn.functional.cross_entropy(v, labels) for v in out.values()) / len(out)
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
        opt.step(); sched.step()
        assert abs(res["loss"].item() - loss.item()) <= 2e-5, (step, res["loss"].item(), loss.item())
        assert abs(res["grad_norm"].item() - norm.item()) <= 2e-4 * max(1.0, norm.item()), (step, res["grad_norm"].item(), norm.item())
    ours.train = tr_train
    worst = max((a - b).abs().max().item() for a, b in zip(ours.parameters(), ref.parameters()))
    assert worst <= 5e-4, worst  # lr / 2: Adam's m / sqrt(v) amplifies rounding differences of near-zero gradients up to the step size
    untouched = [n for (n, a), b in zip(ours.named_parameters(), ref.parameters()) if b.grad is None]
    assert untouched and all("encoder_layer" in n or "score_embeddings" in n or "classifier" in n for n in untouched), untouched


@pytest.mark.gpu
def test_deferred_block_weight_gradients_match_the_per_block_launches(pkg):
    """Round 5: inside a Trainer step the fusion models' blocks queue their weight-gradient products and the end of the backward pass runs
    them as grouped launches of up to 32 products (ops.deferred_block_weight_grads; stlt_ctx_dw_defer / _flush on the trainer's context).  Same gradients as the
    per-block launches to rounding (the grouping changes the stream-K split, i.e. the summation order), products are really queued, a
    weight shared by two blocks (the fusion models' cross-attention, models.py:411-419) accumulates both contributions, and the queue is
    empty afterwards."""
    kw = dict(pkg.synth.model_kwargs(NAME), **EXTRA)
    kw["hidden_dropout_prob"] = 0.0
    c = pkg.synth.CONFIGS[NAME]
    B = 8  # 8 clips x 16 frames = 128 layout rows (a multiple of 32: grouped launches); the 8 x 33 appearance rows keep their own launches
    batch = pkg.synth.make_batch(B, c["T"], c["N"], seed=31)
    batch["appearance_features"] = pkg.synth.make_appearance_features(B, seed=32)
    batch = {k: v.to("cuda") for k, v in batch.items()}
    batch["labels"] = torch.randint(0, c["num_classes"], (B,), generator=torch.Generator().manual_seed(6)).cuda()
    lib = pkg._lib.load()
    import threading
    flats, queued = [], []
    for defer in ("here", "other-thread", False):
        m = pkg.CrossAttentionCentralNetFusion(pkg.MultimodalModelConfig(**kw))
        m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=7))
        m.train(False).to("cuda")
        tr = pkg.train.Trainer(m, "something", learning_rate=1e-3, weight_decay=1e-3, clip_val=5.0, warmup_steps=0, total_steps=1000)
        tr.bound.zero()
        heads = list(m(batch).values())
        grads = [pkg.train.fused_criterion(v, batch["labels"], "something", 1.0 / len(heads))[1] for v in heads]
        tr.bound.accumulating = True
        try:
            if defer:
                with pkg.ops.deferred_block_weight_grads(tr.context):
                    if defer == "here":
                        torch.autograd.backward(heads, grads)
                    else:
                        # torch's autograd engine runs the backward nodes on a thread of its own choosing; the queue follows the context
                        # handle every block call names (a thread-local one was never flushed when the nodes ran elsewhere: round-5 regression)
                        th = threading.Thread(target=lambda: torch.autograd.backward(heads, grads))
                        th.start()
                        th.join()
                    queued.append(tr.context.dw_pending())
            else:
                torch.autograd.backward(heads, grads)
        finally:
            tr.bound.accumulating = False
        assert tr.context.dw_pending() == 0
        torch.cuda.synchronize()
        flats.append(tr.bound.flat.clone())
    assert queued[0] >= 8 and queued[1] == queued[0], queued
    a, a2, b = flats
    assert torch.isfinite(a).all() and a.abs().max().item() > 0
    assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    assert (a2 - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    assert lib.stlt_ctx_dw_defer(tr.context.handle, 7) != 0 and lib.stlt_ctx_dw_defer(tr.context.handle, -1) == 0
    assert lib.stlt_ctx_dw_defer(None, 1) != 0 and lib.stlt_ctx_dw_pending(None) == -1
