"""World-size-2 gloo run (CPU): batch sharding, logits gather and metric reduction of the multi-rank inference path.
The per-rank compute is the CPU oracle standing in for the HIP forward (tests may use the oracle)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG_NAME, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    pkg = importlib.import_module(PKG_NAME)
    D = importlib.import_module(PKG_NAME + ".dist")
    I = importlib.import_module(PKG_NAME + ".infer")
    from oracle import stlt_oracle as O

    r, w = D.init_distributed("gloo")
    assert (r, w) == (rank, world)
    name = "micro"
    c = pkg.synth.CONFIGS[name]
    kw = pkg.synth.model_kwargs(name)
    model = pkg.Stlt(pkg.StltModelConfig(**kw))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=7)
    batches = []
    for i, n in enumerate(n_clips):
        b = pkg.synth.make_batch(n, c["T"], c["N"], seed=100 + i)
        b["labels"] = torch.randint(0, c["num_classes"], (n,), generator=torch.Generator().manual_seed(i))
        batches.append(b)
    fwd = lambda b: O.stlt_forward(sd, b, c["num_attention_heads"])["stlt"]
    res = I.run_inference(None, batches, "cpu", rank, world, forward=fwd, collect_logits=True)
    if rank == 0:
        full = [fwd(b) for b in batches]
        q.put((res["logits"].numpy(), torch.cat(full).numpy(), res["top1_accuracy"], res["top5_accuracy"], res["num_clips"],
               [I.topk_counts(f, b["labels"]).tolist() for f, b in zip(full, batches)]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_clips", [(6, 4), (5, 1, 3)])  # even shards, uneven shards and a batch smaller than the world
def test_two_rank_sharded_inference_matches_single_process(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    logits, full, top1, top5, n, counts = got
    assert n == sum(n_clips)
    assert logits.shape == full.shape
    assert np.abs(logits - full).max() <= 1e-6  # same per-clip math, only the split differs
    t1 = sum(c[0] for c in counts)
    t5 = sum(c[1] for c in counts)
    assert top1 == round(100.0 * t1 / n, 2) and top5 == round(100.0 * t5 / n, 2)


def test_shard_bounds_cover_everything():
    D = importlib.import_module(PKG_NAME + ".dist")
    for n in (0, 1, 7, 8, 1024):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _eval_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    D = importlib.import_module(PKG_NAME + ".dist")
    E = importlib.import_module(PKG_NAME + ".utils.evaluation")
    D.init_distributed("gloo")
    G = np.load(os.path.join(ROOT, "tests", "golden", "evaluation.npz"))
    n = G["sth_labels"].shape[0]
    lo, hi = D.shard_bounds(n, rank, world)
    ev = E.EvaluatorSomething(n, 174, ("stlt", "caf"))
    ev.process({"stlt": torch.from_numpy(G["sth_logits_a"][lo:hi]), "caf": torch.from_numpy(G["sth_logits_b"][lo:hi])}, torch.from_numpy(G["sth_labels"][lo:hi]))
    m = ev.evaluate()
    keep = [j for j in range(157) if j != 11]
    lg, gt = G["ag_logits"][:119, keep], G["ag_truths"][:119, keep]  # 119 clips: uneven shards
    lo, hi = D.shard_bounds(119, rank, world)
    ag = E.EvaluatorActionGenome(119, len(keep), ("stlt",))
    ag.process({"stlt": torch.from_numpy(lg[lo:hi])}, torch.from_numpy(gt[lo:hi]))
    got = ag.evaluate()["map"]
    if rank == 0:
        one = E.EvaluatorActionGenome(119, len(keep), ("stlt",))
        one.predictions = torch.from_numpy(lg).sigmoid().double()
        one.ground_truths = torch.from_numpy(gt).double()
        one.index = 119
        ref = float(E.charades_map(one.predictions, one.ground_truths)[0])
        q.put((m, got, ref))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_evaluators_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    m, got, ref = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    G = np.load(os.path.join(ROOT, "tests", "golden", "evaluation.npz"))
    assert [m["stlt_top1_accuracy"], m["stlt_top5_accuracy"], m["caf_top1_accuracy"], m["caf_top5_accuracy"]] == G["sth_metrics"].tolist()
    assert abs(got - ref) < 1e-12
