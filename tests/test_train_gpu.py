"""Training step parity: the native forward/backward against torch autograd on the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden_case
from oracle import stlt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _oracle_grads(sd, batch, H, labels, dtype=torch.float64):
    leaves = {k: (v.detach().to(dtype).requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    b = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in batch.items()}
    logits = O.stlt_forward(leaves, b, H, dtype=dtype)["stlt"]
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    return loss.item(), logits.detach(), {k: v.grad for k, v in leaves.items() if v.is_floating_point()}


@pytest.mark.parametrize("skip_padding", [False, True])
@pytest.mark.parametrize("name,B,with_scores", [("cfg1", 3, False), ("cfg1", 5, True), ("cfg2", 2, False), ("cfg4", 2, True)])
def test_gradients_match_oracle_autograd(pkg, name, B, with_scores, skip_padding):
    c = pkg.synth.CONFIGS[name]
    H = c["num_attention_heads"]
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=31, gain=1.5)
    m.load_state_dict(sd)
    m.train(True)  # hidden_dropout_prob = 0 in model_kwargs
    m.to(DEV)
    m.backbone.skip_padding = skip_padding  # forward + reverse sweep on the real tokens / frames only
    batch = pkg.synth.make_batch(B, c["T"], c["N"], dataset=c["dataset"], seed=77, with_scores=with_scores)
    labels = torch.randint(0, c["num_classes"], (B,), generator=torch.Generator().manual_seed(5))
    ref_loss, ref_logits, ref_g = _oracle_grads(sd, batch, H, labels)

    out = m({k: v.to(DEV) for k, v in batch.items()})["stlt"]
    assert out.requires_grad
    assert (out.detach().cpu().double() - ref_logits).abs().max().item() <= 1e-4
    loss = F.cross_entropy(out, labels.to(DEV))
    assert abs(loss.item() - ref_loss) <= 1e-5
    loss.backward()

    worst = ("", 0.0)
    n_checked = 0
    for k, p in m.named_parameters():
        g_ref = ref_g[k]
        dead = "encoder_layer." in k or ("score_embeddings" in k and not with_scores)
        if dead:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k
            assert g_ref is None or g_ref.abs().max().item() == 0.0, k
            continue
        assert p.grad is not None, k
        scale = max(g_ref.abs().max().item(), 1e-6)
        err = (p.grad.cpu().double() - g_ref).abs().max().item() / scale
        n_checked += 1
        if err > worst[1]:
            worst = (k, err)
        assert err <= 2e-4, f"{k}: relative grad error {err:.2e} (|g|max={scale:.2e})"
    print(f"{name} B={B} skip_padding={skip_padding}: {n_checked} gradients checked, worst {worst[0]} rel err {worst[1]:.2e}")
    # padding_idx rows get no gradient (models.py:22,91)
    assert m.backbone.frames_embeddings.frame_type_embedding.weight.grad[0].abs().max().item() == 0.0
    assert m.backbone.frames_embeddings.layout_embedding.category_box_embeddings.category_embeddings.weight.grad[0].abs().max().item() == 0.0


def test_backward_is_bitwise_reproducible_and_accumulates(pkg):
    name = "cfg1"
    c = pkg.synth.CONFIGS[name]
    sd, batch, z, meta = golden_case(name)
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
    m.load_state_dict(sd)
    m.train(True).to(DEV)
    b = {k: v.to(DEV) for k, v in batch.items()}
    labels = torch.arange(batch["categories"].shape[0], device=DEV) % c["num_classes"]
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        F.cross_entropy(m(b)["stlt"], labels).backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert all(torch.equal(grads[0][k], grads[1][k]) for k in grads[0])  # slab reductions, no float atomics
    F.cross_entropy(m(b)["stlt"], labels).backward()  # second backward without zero_grad: autograd accumulates
    k0 = "prediction_head.fc2.weight"
    assert (dict(m.named_parameters())[k0].grad - 2 * grads[0][k0]).abs().max().item() <= 1e-6 * grads[0][k0].abs().max().item() + 1e-9


def test_frozen_backbone_trains_head_only(pkg, tmp_path):
    kw = pkg.synth.model_kwargs("cfg1")
    c = pkg.synth.CONFIGS["cfg1"]
    bb = pkg.StltBackbone(pkg.StltModelConfig(**kw))
    path = str(tmp_path / "bb.pt")
    torch.save(bb.state_dict(), path)
    m = pkg.Stlt(pkg.StltModelConfig(**kw, load_backbone_path=path, freeze_backbone=True))
    m.train(True).to(DEV)
    batch = {k: v.to(DEV) for k, v in pkg.synth.make_batch(4, c["T"], c["N"], seed=3).items()}
    F.cross_entropy(m(batch)["stlt"], torch.tensor([1, 2, 3, 4], device=DEV)).backward()
    assert all(p.grad is None for p in m.backbone.parameters())
    assert all(p.grad is not None and p.grad.abs().max().item() > 0 for p in m.prediction_head.parameters())


def test_three_training_steps_match_reference_golden(pkg):
    """Reference train step (train.py:119-135 with parser defaults, warm-up 2 of 10 steps) captured by
    tools/gen_golden_train.py: loss, clip_grad_norm_ value and parameter slices after each step."""
    from conftest import GOLDEN
    import os
    z = np.load(os.path.join(GOLDEN, "train_cfg1.npz"))
    name = "cfg1"
    c = pkg.synth.CONFIGS[name]
    B = int(z["batch"][0])
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234)
    m.load_state_dict(sd)
    m.to(DEV)
    tr = pkg.train.Trainer(m, "something", learning_rate=5e-5, weight_decay=1e-3, clip_val=5.0, warmup_steps=2, total_steps=10)
    watch = ["prediction_head.fc2.bias", "prediction_head.fc2.weight",
             "backbone.frames_embeddings.layout_embedding.category_box_embeddings.category_embeddings.weight",
             "backbone.frames_embeddings.layout_embedding.transformer.layers.0.self_attn.in_proj_weight",
             "backbone.transformer.layers.7.linear2.weight", "backbone.frames_embeddings.position_embeddings.weight",
             "backbone.transformer.layers.3.norm1.weight"]
    params = dict(m.named_parameters())
    for s in range(int(z["steps"][0])):
        batch = pkg.synth.make_batch(B, c["T"], c["N"], seed=500 + s)
        batch["labels"] = torch.randint(0, c["num_classes"], (B,), generator=torch.Generator().manual_seed(900 + s))
        out = tr.step({k: v.to(DEV) for k, v in batch.items()})
        assert abs(float(out["loss"]) - float(z[f"loss{s}"][0])) <= 2e-5, s
        assert abs(float(out["grad_norm"]) - float(z[f"gnorm{s}"][0])) <= 2e-4 * float(z[f"gnorm{s}"][0]), s
        for i, k in enumerate(watch):
            got = params[k].detach().reshape(-1)[:64].cpu().numpy()
            assert np.abs(got - z[f"p{s}_{i}"]).max() <= 2.5e-5, (s, k)  # < lr/2 (Adam amplifies last-bit gradient noise)
    assert sum(1 for p in m.parameters() if p.grad is None) == int(z["n_grad_none"][0])  # 14 dead / unused parameters


@pytest.mark.parametrize("name,B,p", [("cfg1", 3, 0.1), ("cfg1", 2, 0.5)])
def test_dropout_forward_and_gradients_match_masked_oracle(pkg, name, B, p):
    """hidden_dropout_prob > 0 in train mode: the six dropout sites of the reference (both embedding outputs, attention
    probabilities, dropout1, FFN dropout, dropout2) with the build's counter-based masks, against the oracle applying
    the same masks under torch autograd."""
    c = pkg.synth.CONFIGS[name]
    H = c["num_attention_heads"]
    kw = dict(pkg.synth.model_kwargs(name), hidden_dropout_prob=p)
    m = pkg.Stlt(pkg.StltModelConfig(**kw))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=41, gain=1.5)
    m.load_state_dict(sd)
    m.train(True).to(DEV)
    seed = 123456789
    m._dropout_seed_override = seed
    batch = pkg.synth.make_batch(B, c["T"], c["N"], seed=9)
    labels = torch.randint(0, c["num_classes"], (B,), generator=torch.Generator().manual_seed(1))
    drop = O.Dropout(p, seed)
    leaves = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    ref_logits = O.stlt_forward(leaves, b64, H, dtype=torch.float64, drop=drop)["stlt"]
    F.cross_entropy(ref_logits, labels).backward()

    out = m({k: v.to(DEV) for k, v in batch.items()})["stlt"]
    assert (out.detach().cpu().double() - ref_logits.detach()).abs().max().item() <= 2e-4
    F.cross_entropy(out, labels.to(DEV)).backward()
    for k, prm in m.named_parameters():
        if "encoder_layer." in k or "score_embeddings" in k:
            continue
        g_ref = leaves[k].grad
        scale = max(g_ref.abs().max().item(), 1e-6)
        assert (prm.grad.cpu().double() - g_ref).abs().max().item() / scale <= 5e-4, k
    # dropout really is on: a different seed changes the logits, eval mode ignores it, and the rate is right
    m._dropout_seed_override = seed + 1
    with torch.enable_grad():
        other = m({k: v.to(DEV) for k, v in batch.items()})["stlt"]
    assert (other - out).abs().max().item() > 1e-3
    keep = O.dropout_keep(p, seed, 17, np.arange(200000, dtype=np.uint64))
    assert abs(1.0 - keep.mean() - p) < 0.01


def test_skip_padding_training_edge_layouts_and_alternation(pkg):
    """Ragged row counts change from step to step and the padded schedule can follow a ragged step on the same tape /
    scratch buffers: gradients must match the padded schedule's every time (stale rows past the row count must not leak)."""
    name = "cfg1"
    c = pkg.synth.CONFIGS[name]
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=8, gain=1.5))
    m.train(True).to(DEV)
    B = 6
    batches = [pkg.synth.make_batch(B, c["T"], c["N"], seed=41, dense=True),          # nothing to skip
               pkg.synth.make_batch(B, c["T"], c["N"], seed=42, min_len=2),            # short clips
               pkg.synth.make_batch(B, c["T"], c["N"], seed=43),
               pkg.synth.make_batch(B, c["T"], c["N"], seed=44, min_len=2)]
    empty = pkg.synth.make_batch(B, c["T"], c["N"], seed=45)
    empty["categories"][:, :, 1:] = 0
    empty["boxes"][:, :, 1:] = 0
    empty["src_key_padding_mask_boxes"] = empty["categories"] == 0                       # CLS-only frames
    batches.append(empty)
    labels = torch.randint(0, c["num_classes"], (B,), generator=torch.Generator().manual_seed(5)).to(DEV)

    def grads(batch, skip):
        m.backbone.skip_padding = skip
        m.zero_grad(set_to_none=True)
        out = m({k: v.to(DEV) for k, v in batch.items()})["stlt"]
        F.cross_entropy(out, labels).backward()
        return out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    for batch in batches:
        ref_out, ref = grads(batch, False)
        out, got = grads(batch, True)
        assert (out - ref_out).abs().max().item() <= 2e-5
        assert set(got) == set(ref)
        for k in ref:
            scale = max(ref[k].abs().max().item(), 1e-6)
            assert (got[k] - ref[k]).abs().max().item() / scale <= 2e-4, k
        again_out, again = grads(batch, True)
        assert torch.equal(out, again_out) and all(torch.equal(got[k], again[k]) for k in got)  # reproducible


def test_skip_padding_training_with_dropout_is_seeded_and_finite(pkg):
    name = "cfg1"
    c = pkg.synth.CONFIGS[name]
    kw = pkg.synth.model_kwargs(name)
    kw["hidden_dropout_prob"] = 0.1
    m = pkg.Stlt(pkg.StltModelConfig(**kw))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=8))
    m.train(True).to(DEV)
    m.backbone.skip_padding = True
    batch = {k: v.to(DEV) for k, v in pkg.synth.make_batch(4, c["T"], c["N"], seed=3).items()}
    labels = torch.randint(0, c["num_classes"], (4,), generator=torch.Generator().manual_seed(5)).to(DEV)

    def run(seed):
        torch.manual_seed(seed)
        m.zero_grad(set_to_none=True)
        out = m(batch)["stlt"]
        F.cross_entropy(out, labels).backward()
        g = m.backbone.transformer.layers[0].linear1.weight.grad.detach().clone()
        return out.detach().clone(), g

    a, ga = run(1)
    b, gb = run(1)
    c2, gc = run(2)
    assert torch.isfinite(a).all() and torch.isfinite(ga).all()
    assert torch.equal(a, b) and torch.equal(ga, gb)          # same seed -> same masks in forward and backward
    assert not torch.equal(a, c2)                             # another seed -> another mask
    m.train(False)
    with torch.no_grad():
        e = m(batch)["stlt"]
    assert (e - a).abs().max().item() > 1e-4                  # dropout really was applied in training mode
