"""World-size-2 gloo run (CPU) of the data-parallel training step: shard the global batch, average gradients with one
all-reduce, clip, AdamW — must reproduce the single-process step on the global batch.  The per-rank forward/backward
is the CPU oracle under torch autograd standing in for the HIP step (tests may use the oracle)."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import PKG_NAME, ROOT


class _OracleStlt(torch.nn.Module):
    """nn.Parameters named like the drop-in's state dict; forward = oracle restatement (differentiable torch ops)."""

    def __init__(self, sd, num_heads):
        super().__init__()
        self.names = [k for k, v in sd.items() if v.is_floating_point()]
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(sd[k].clone()) for k in self.names])
        self.extra = {k: v for k, v in sd.items() if not v.is_floating_point()}
        self.H = num_heads

    def named_parameters(self, *a, **kw):  # reference names decide the weight-decay groups
        return list(zip(self.names, self.ps))

    def forward(self, batch):
        from oracle import stlt_oracle as O
        sd = dict(zip(self.names, self.ps))
        sd.update(self.extra)
        return O.stlt_forward(sd, batch, self.H)


def _setup(pkg, seed_batch, n):
    c = pkg.synth.CONFIGS["micro"]
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("micro")))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=11)
    batch = pkg.synth.make_batch(n, c["T"], c["N"], seed=seed_batch)
    batch["labels"] = torch.randint(0, c["num_classes"], (n,), generator=torch.Generator().manual_seed(seed_batch))
    return c, sd, batch


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    pkg = importlib.import_module(PKG_NAME)
    r, w = pkg.dist.init_distributed("gloo")
    c, sd, _ = _setup(pkg, 0, 8)
    model = _OracleStlt(sd, c["num_attention_heads"])
    tr = pkg.train.Trainer(model, "something", warmup_steps=1, total_steps=5, rank=r, world=w)
    batches = [_setup(pkg, 20 + s, 8)[2] for s in range(3)]
    log = tr.fit(batches, "cpu")
    if rank == 0:
        q.put((log, {k: p.detach().numpy().copy() for k, p in model.named_parameters()}))
    # the epoch shell over two ranks: sharded optimisation steps, sharded validation into the evaluator (counters summed over the group in
    # evaluate()), is_best on every rank, files written by rank 0 only
    import tempfile
    hist, files = _fit_epochs(pkg, r, w, tempfile.mkdtemp() if rank == 0 else "/nonexistent-dir-rank1")
    if rank == 0:
        q.put((hist, files))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _fit_epochs(pkg, rank, world, out_dir):
    task = pkg.synth.FIT_TASK
    c = pkg.synth.CONFIGS[task["config"]]
    shapes = {k: tuple(v.shape) for k, v in pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(task["config"]))).state_dict().items()}
    model = _OracleStlt(pkg.synth.make_state_dict(shapes, seed=task["weight_seed"]), c["num_attention_heads"])
    nb = task["train_batches"]
    tr = pkg.train.Trainer(model, "something", learning_rate=task["lr"], weight_decay=task["weight_decay"], clip_val=task["clip_val"],
                           warmup_steps=task["warmup_epochs"] * nb, total_steps=task["epochs"] * nb, rank=rank, world=world)
    val = [pkg.synth.fit_batch("val", 0, i) for i in range(task["val_batches"])]
    ev = pkg.evaluators_factory["something"](sum(b["labels"].shape[0] for b in val), c["num_classes"], ("stlt",))
    path = os.path.join(out_dir, "model.pt")
    hist = tr.fit_epochs(lambda e: [pkg.synth.fit_batch("train", e, i) for i in range(nb)], val, ev, task["epochs"], "cpu", save_model_path=path)
    return hist, (os.path.exists(path), [rec["saved"] for rec in hist])


def test_two_rank_training_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    log2, params2 = q.get(timeout=300)
    hist2, (wrote2, saved2) = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pkg = importlib.import_module(PKG_NAME)
    c, sd, _ = _setup(pkg, 0, 8)
    model = _OracleStlt(sd, c["num_attention_heads"])
    tr = pkg.train.Trainer(model, "something", warmup_steps=1, total_steps=5)
    log1 = tr.fit([_setup(pkg, 20 + s, 8)[2] for s in range(3)], "cpu")
    for a, b in zip(log1, log2):
        assert abs(a["loss"] - b["loss"]) <= 1e-5
        assert abs(a["grad_norm"] - b["grad_norm"]) <= 1e-4 * a["grad_norm"]
    for k, p in model.named_parameters():
        assert np.abs(p.detach().numpy() - params2[k]).max() <= 2.5e-5, k  # < lr/2: Adam turns last-bit gradient noise into O(lr) steps


    # the epoch shell: two ranks give the reference loop's metrics and best-epoch pattern (tests/golden/fit_micro.npz), rank 0 wrote the file
    z = np.load(os.path.join(ROOT, "tests", "golden", "fit_micro.npz"))
    assert wrote2 and [bool(x) for x in saved2] == [bool(x) for x in z["saved"]]
    for e, rec in enumerate(hist2):
        assert rec["is_best"] == bool(z["saved"][e])
        assert (rec["metrics"]["stlt_top1_accuracy"], rec["metrics"]["stlt_top5_accuracy"]) == tuple(z[f"metrics{e}"])
        assert abs(float(np.mean([s["loss"] for s in rec["steps"]])) - float(z[f"mean_loss{e}"][0])) <= 2e-4


def test_weight_decay_groups_follow_reference_rule():
    pkg = importlib.import_module(PKG_NAME)
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("cfg1")))
    groups = pkg.train.add_weight_decay(m, 1e-3)
    assert groups[0]["weight_decay"] == 0.0 and groups[1]["weight_decay"] == 1e-3
    assert len(groups[0]["params"]) == 114 and len(groups[1]["params"]) == 59  # probed on the reference (173 params)
    assert all(p.dim() == 1 for p in groups[0]["params"])
    sched_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch = pkg.train.linear_schedule_with_warmup(sched_opt, 2, 10)
    lrs = []
    for _ in range(11):
        lrs.append(sched_opt.param_groups[0]["lr"])
        sched_opt.step(); sch.step()
    assert lrs[:4] == [0.0, 0.5, 1.0, 0.875] and lrs[10] == 0.0
