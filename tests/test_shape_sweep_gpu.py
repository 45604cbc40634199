"""Randomised layout shapes (frames T, object slots N, batch B are data shapes, not parameters: one cfg1-sized model
serves them all) against the CPU oracle: padded and skip-padding forward, and the training gradients of both
schedules.  Includes the corner shapes: a single object slot (CLS only), two frames, slot counts above the 32-row
attention tile, and frame / slot counts above 64, where the attention backward switches to its streamed long-sequence
kernel (up to 256, the size of the position table)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import stlt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(1, 2, 1), (3, 2, 2), (2, 5, 1), (4, 9, 3), (2, 17, 12), (3, 16, 33), (1, 31, 8), (2, 33, 5), (5, 40, 2), (2, 64, 9),
          (1, 65, 4), (2, 100, 3), (1, 6, 70), (1, 256, 2)]


@pytest.fixture(scope="module")
def model(pkg):
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("cfg1")))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=77, gain=1.5)
    m.load_state_dict(sd)
    return m.to(DEV), sd


@pytest.mark.parametrize("B,T,N", SHAPES)
def test_forward_padded_and_skip_padding_match_oracle(pkg, model, B, T, N):
    m, sd = model
    m.train(False)
    batch = pkg.synth.make_batch(B, T, N, seed=1000 * T + N, min_len=2)
    ref = O.stlt_forward(sd, batch, 4)["stlt"]
    dev = {k: v.to(DEV) for k, v in batch.items()}
    with torch.no_grad():
        for skip in (False, True):
            m.backbone.skip_padding = skip
            got = m(dev)["stlt"].cpu()
            assert (got - ref).abs().max().item() <= 1e-4, (skip, (got - ref).abs().max().item())
    m.backbone.skip_padding = False


@pytest.mark.parametrize("B,T,N", SHAPES)
def test_training_gradients_match_oracle_both_schedules(pkg, model, B, T, N):
    m, sd = model
    m.train(True)  # dropout 0 in cfg1's kwargs
    batch = pkg.synth.make_batch(B, T, N, seed=2000 * T + N, min_len=2)
    labels = torch.randint(0, 174, (B,), generator=torch.Generator().manual_seed(T * 7 + N))
    leaves = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    F.cross_entropy(O.stlt_forward(leaves, b64, 4, dtype=torch.float64)["stlt"], labels).backward()
    dev = {k: v.to(DEV) for k, v in batch.items()}
    for skip in (False, True):
        m.backbone.skip_padding = skip
        m.zero_grad(set_to_none=True)
        F.cross_entropy(m(dev)["stlt"], labels.to(DEV)).backward()
        for k, p in m.named_parameters():
            g_ref = leaves[k].grad
            if p.grad is None:
                assert g_ref is None or g_ref.abs().max().item() == 0.0, k
                continue
            scale = max(g_ref.abs().max().item(), 1e-6)
            assert (p.grad.cpu().double() - g_ref).abs().max().item() / scale <= 3e-4, (skip, k)
    m.backbone.skip_padding = False
    m.train(False)


def test_long_sequence_training_with_dropout_is_seeded(pkg):
    """The streamed attention backward recomputes the same dropout masks as the forward (T = 80 > 64)."""
    kw = pkg.synth.model_kwargs("cfg1")
    kw["hidden_dropout_prob"] = 0.2
    m = pkg.Stlt(pkg.StltModelConfig(**kw))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=77))
    m.train(True).to(DEV)
    batch = {k: v.to(DEV) for k, v in pkg.synth.make_batch(2, 80, 3, seed=5).items()}
    labels = torch.tensor([3, 9], device=DEV)

    def run(seed):
        torch.manual_seed(seed)
        m.zero_grad(set_to_none=True)
        out = m(batch)["stlt"]
        F.cross_entropy(out, labels).backward()
        return out.detach().clone(), m.backbone.transformer.layers[0].self_attn.in_proj_weight.grad.detach().clone()

    a, ga = run(1)
    b, gb = run(1)
    assert torch.isfinite(ga).all() and torch.equal(a, b) and torch.equal(ga, gb)
    # finite-difference check of one direction through the whole stochastic graph (same masks: same seed)
    w = m.backbone.transformer.layers[0].self_attn.in_proj_weight
    direction = torch.randn_like(w) * 3e-3
    with torch.no_grad():
        w.add_(direction)
    torch.manual_seed(1)
    lp = F.cross_entropy(m(batch)["stlt"], labels).item()
    with torch.no_grad():
        w.sub_(2 * direction)
    torch.manual_seed(1)
    lm = F.cross_entropy(m(batch)["stlt"], labels).item()
    with torch.no_grad():
        w.add_(direction)
    fd = (lp - lm) / 2
    an = (ga * direction).sum().item()
    assert abs(fd - an) <= 1e-2 * max(abs(an), 1e-3) + 2e-5, (fd, an)
    m.train(False)


@pytest.mark.parametrize("d,H", [(64, 1), (128, 2), (512, 8), (1024, 16)])
def test_other_hidden_sizes_forward_and_gradients(pkg, d, H):
    """The kernels are written for head dim 64 and any hidden size that is a multiple of it (up to 2048): released
    checkpoints use 768 / 12, the tests mostly 256 / 4 — cover a few more."""
    kw = dict(pkg.synth.model_kwargs("cfg1"), hidden_size=d, num_attention_heads=H, num_spatial_layers=2, num_temporal_layers=2)
    m = pkg.Stlt(pkg.StltModelConfig(**kw))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=13, gain=1.5)
    m.load_state_dict(sd)
    m.to(DEV)
    batch = pkg.synth.make_batch(3, 9, 5, seed=6, min_len=2)
    dev = {k: v.to(DEV) for k, v in batch.items()}
    ref = O.stlt_forward(sd, batch, H)["stlt"]
    m.train(False)
    with torch.no_grad():
        for skip in (False, True):
            m.backbone.skip_padding = skip
            assert (m(dev)["stlt"].cpu() - ref).abs().max().item() <= 1e-4, skip
    m.backbone.skip_padding = False
    m.train(True)
    labels = torch.tensor([1, 2, 3])
    leaves = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    F.cross_entropy(O.stlt_forward(leaves, b64, H, dtype=torch.float64)["stlt"], labels).backward()
    F.cross_entropy(m(dev)["stlt"], labels.to(DEV)).backward()
    for k, prm in m.named_parameters():
        if prm.grad is None:
            continue
        g_ref = leaves[k].grad
        scale = max(g_ref.abs().max().item(), 1e-6)
        assert (prm.grad.cpu().double() - g_ref).abs().max().item() / scale <= 3e-4, k
