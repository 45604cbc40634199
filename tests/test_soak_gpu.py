"""Soak: the same inputs must give the same bits every time, across interleaved shapes, schedules and fresh allocations
(an intermittent cold-cache bug in an early GEMM epilogue showed up exactly this way: 1 run in ~100 differed)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_interleaved_forwards_are_bitwise_stable(pkg):
    name = "cfg1"
    c = pkg.synth.CONFIGS[name]
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3, gain=1.5))
    m = m.train(False).to(DEV)
    cases = []
    for i, (B, T, N) in enumerate([(8, 16, 4), (3, 9, 7), (64, 16, 4), (1, 33, 2)]):
        batch = {k: v.to(DEV) for k, v in pkg.synth.make_batch(B, T, N, seed=40 + i, min_len=2).items()}
        for skip in (False, True):
            cases.append((batch, skip))
    first = {}
    junk = []
    with torch.no_grad():
        for it in range(40):
            for ci, (batch, skip) in enumerate(cases):
                m.backbone.skip_padding = skip
                out = m(batch)["stlt"]
                if ci not in first:
                    first[ci] = out.clone()
                else:
                    assert torch.equal(out, first[ci]), (it, ci)
            junk.append(torch.empty(1 + 37 * it, 1024, device=DEV))  # shift later allocations around
            if it % 10 == 9:
                m.backbone._ws.buf = None  # force a fresh workspace at a new address
    m.backbone.skip_padding = False


def test_linear_is_bitwise_stable_across_fresh_buffers(pkg):
    g = torch.Generator().manual_seed(1)
    shapes = [(1, 64, 32, 1), (300, 768, 96, 0), (2048, 768, 768, 0), (5000, 3072, 768, 1), (1000, 174, 256, 0)]
    data = []
    for M, N, K, act in shapes:
        x = (torch.rand(M, K, generator=g) * 2 - 1).to(DEV)
        w = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(DEV)
        b = torch.rand(N, generator=g).to(DEV)
        data.append((x, w, b, act))
    ref = {}
    keep = []
    for it in range(60):
        with_scratch = it % 2 == 1
        for si, (x, w, b, act) in enumerate(data):
            xx, ww, bb = x.clone(), w.clone(), b.clone()  # fresh (cold) buffers every time
            if with_scratch:
                with pkg.ops.gemm_scratch():
                    y = pkg.ops.linear(xx, ww, bb, act=act)
            else:
                y = pkg.ops.linear(xx, ww, bb, act=act)
            key = (si, with_scratch)
            if key not in ref:
                ref[key] = y.clone()
            else:
                assert torch.equal(y, ref[key]), (it, key)
        keep.append(torch.empty(3 + 11 * it, 512, device=DEV))
