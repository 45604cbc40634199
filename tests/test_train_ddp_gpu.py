"""Two data-parallel ranks (gloo rendezvous, both on cuda:0) running the native training step with the fused optimiser
and the overlapped two-slice gradient all-reduce, against one process stepping on the global batch."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG_NAME, ROOT

pytestmark = pytest.mark.gpu
NAME, STEPS, GLOBAL_B = "cfg1", 3, 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(pkg):
    c = pkg.synth.CONFIGS[NAME]
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(NAME)))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=21, gain=1.5))
    batches = []
    for s in range(STEPS):
        b = pkg.synth.make_batch(GLOBAL_B, c["T"], c["N"], seed=300 + s)
        b["labels"] = torch.randint(0, c["num_classes"], (GLOBAL_B,), generator=torch.Generator().manual_seed(s))
        batches.append(b)
    return m.to("cuda"), batches


def _worker(rank, world, port, skip_padding, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    pkg = importlib.import_module(PKG_NAME)
    r, w = pkg.dist.init_distributed("gloo")
    torch.cuda.set_device(0)
    m, batches = _make(pkg)
    m.backbone.skip_padding = skip_padding
    tr = pkg.train.Trainer(m, "something", learning_rate=1e-3, warmup_steps=1, total_steps=10, rank=r, world=w)
    assert tr.fused
    log = tr.fit(batches, "cuda")
    assert tr._comm_stream is not None  # the slices went through the side stream
    if rank == 0:
        q.put((log, {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}))  # numpy: no fd passing
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("skip_padding", [False, True])
def test_two_ranks_match_single_process_global_batch(pkg, skip_padding):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, skip_padding, q)) for r in range(2)]
    for p in procs:
        p.start()
    log2, sd2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m, batches = _make(pkg)
    m.backbone.skip_padding = skip_padding
    tr = pkg.train.Trainer(m, "something", learning_rate=1e-3, warmup_steps=1, total_steps=10)
    log1 = tr.fit(batches, "cuda")
    for a, b in zip(log1, log2):
        assert abs(a["loss"] - b["loss"]) <= 1e-5
        assert abs(a["grad_norm"] - b["grad_norm"]) <= 1e-4 * max(a["grad_norm"], 1.0)
    sd1 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for k in sd1:
        if not sd1[k].is_floating_point():
            continue
        a, b = sd1[k], torch.from_numpy(sd2[k])
        if k.endswith("in_proj_bias"):  # the key-bias third only ever sees rounding noise (see test_train_gpu.py)
            d = a.numel() // 3
            a, b = torch.cat([a[:d], a[2 * d:]]), torch.cat([b[:d], b[2 * d:]])
        assert (a - b).abs().max().item() <= 2e-5, k
