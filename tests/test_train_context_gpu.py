"""Training-loop contexts (include/stlt_hip.h: stlt_ctx): what a training loop leaves inside the library between calls — transposed weight
copies, deferred block weight gradients, the side stream of its reverse sweeps — belongs to its own handle.  Two trainers in one process,
alternating steps or stepping from two host threads, give the parameters each gives alone, bit for bit; a plain autograd backward of another
model in the middle of a trainer's step reads no copy and queues nothing."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
STEPS = 3


def _stlt(pkg):
    kw = dict(pkg.synth.model_kwargs("cfg1"), hidden_dropout_prob=0.0)
    m = pkg.Stlt(pkg.StltModelConfig(**kw))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234))
    return m.to(DEV)


def _cacnf(pkg):
    kw = dict(pkg.synth.model_kwargs("cfg1"), appearance_num_frames=32, num_appearance_layers=2, num_fusion_layers=2, hidden_dropout_prob=0.0)
    m = pkg.CrossAttentionCentralNetFusion(pkg.MultimodalModelConfig(**kw))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=7))
    return m.to(DEV)


def _batches(pkg, fusion, n):
    c = pkg.synth.CONFIGS["cfg1"]
    out = []
    for s in range(STEPS):
        b = pkg.synth.make_batch(n, c["T"], c["N"], seed=300 + s + (50 if fusion else 0))
        if fusion:
            b["appearance_features"] = pkg.synth.make_appearance_features(n, seed=60 + s)
        b["labels"] = torch.randint(0, c["num_classes"], (n,), generator=torch.Generator().manual_seed(400 + s))
        out.append({k: v.to(DEV) for k, v in b.items()})
    return out


def _trainer(pkg, model):
    return pkg.train.Trainer(model, "something", learning_rate=1e-3, weight_decay=1e-3, clip_val=5.0, warmup_steps=0, total_steps=100)


def _alone(pkg, make, fusion, n):
    # the fusion models' appearance encoder always trains with dropout 0.1 (models.py:239-246): its per-call seeds come from torch's CPU
    # generator, so every run that is compared starts that generator at the same point (the STLT model, dropout 0, draws nothing)
    m = make(pkg)
    tr = _trainer(pkg, m)
    torch.manual_seed(0)  # after the constructors (parameter initialisation draws from the same generator)
    logs = [tr.step(b) for b in _batches(pkg, fusion, n)]
    torch.cuda.synchronize()
    return [p.detach().clone() for p in m.parameters()], [float(o["loss"]) for o in logs]


def _same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.fixture(scope="module")
def alone(pkg):
    return {"stlt": _alone(pkg, _stlt, False, 32), "cacnf": _alone(pkg, _cacnf, True, 8)}


def test_a_trainer_is_bitwise_reproducible(pkg, alone):
    """The premise of the two tests below."""
    assert _same(alone["stlt"][0], _alone(pkg, _stlt, False, 32)[0])
    assert _same(alone["cacnf"][0], _alone(pkg, _cacnf, True, 8)[0])


def test_two_trainers_alternating_steps_in_one_process(pkg, alone):
    ms, mc = _stlt(pkg), _cacnf(pkg)
    ts, tc = _trainer(pkg, ms), _trainer(pkg, mc)
    assert ts.context is not tc.context and ts.context.handle != tc.context.handle
    bs, bc = _batches(pkg, False, 32), _batches(pkg, True, 8)
    torch.manual_seed(0)
    served = []
    for s in range(STEPS):
        ts.step(bs[s])
        tc.step(bc[s])
        served.append((ts.context.wt_hits(), tc.context.wt_hits()))
    torch.cuda.synchronize()
    assert _same([p.detach() for p in ms.parameters()], alone["stlt"][0])
    assert _same([p.detach() for p in mc.parameters()], alone["cacnf"][0])
    if ts.transposed is not None:  # both loops' input-gradient products were served from their OWN copies
        assert served[-1][0] > 0 and served[-1][1] > 0
    assert ts.context.dw_pending() == 0 and tc.context.dw_pending() == 0


def test_two_trainers_stepping_from_two_host_threads(pkg, alone):
    """One trainer per thread, both on the process's default stream of cuda:0: the host-side interleaving is arbitrary (the reverse sweeps'
    side streams, the deferred queues and the weight copies are per context), the results are each trainer's own."""
    ms, mc = _stlt(pkg), _cacnf(pkg)
    ts, tc = _trainer(pkg, ms), _trainer(pkg, mc)
    bs, bc = _batches(pkg, False, 32), _batches(pkg, True, 8)
    torch.manual_seed(0)
    errs = []

    def run(tr, batches):
        try:
            torch.cuda.set_device(0)
            for b in batches:
                tr.step(b)
        except Exception as exc:  # surfaced below
            errs.append(exc)

    th = [threading.Thread(target=run, args=(ts, bs)), threading.Thread(target=run, args=(tc, bc))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert _same([p.detach() for p in ms.parameters()], alone["stlt"][0])
    assert _same([p.detach() for p in mc.parameters()], alone["cacnf"][0])


def test_a_plain_backward_between_refresh_and_clear_reads_no_copy_and_queues_nothing(pkg):
    """A third model's ordinary autograd backward in the middle of a trainer's step (a teacher's gradient penalty, an evaluation hook): it
    names no context — even on the trainer's OWN weights — so it is never served from the step's copies and nothing lands in its queue."""
    mc = _cacnf(pkg)
    tc = _trainer(pkg, mc)
    third = _stlt(pkg)
    b3 = _batches(pkg, False, 16)[0]
    bc = _batches(pkg, True, 8)[0]
    lib = pkg._lib.load()
    if tc.transposed is None:
        pytest.skip("STLT_TRAIN_WT=0")
    tc.transposed.refresh()
    try:
        pkg._lib.check(lib.stlt_ctx_dw_defer(tc.context.handle, 1), "stlt_ctx_dw_defer")
        h0 = tc.context.wt_hits()
        # (a) another model, plain autograd
        third.train(True)
        torch.nn.functional.cross_entropy(third(b3)["stlt"], b3["labels"]).backward()
        # (b) the trainer's own model under plain autograd (its parameters are bound to the trainer's flat buffer, but no step is running)
        mc.train(True)
        out = mc(bc)
        sum(torch.nn.functional.cross_entropy(v, bc["labels"]) for v in out.values()).backward()
        torch.cuda.synchronize()
        assert tc.context.wt_hits() == h0 and tc.context.dw_pending() == 0
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in third.prediction_head.parameters())
    finally:
        lib.stlt_ctx_dw_defer(tc.context.handle, -1)
        tc.transposed.clear()
    # and the trainer still steps
    r = tc.step(bc)
    assert float(r["loss"]) == float(r["loss"]) and tc.context.wt_hits() > h0
