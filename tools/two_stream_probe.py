#!/usr/bin/env python3
"""Probe (GPU box only): does running two half-batches of the forward on two HIP streams fill the idle tails of the
persistent GEMM launches?  Compares one 1024-clip forward per step with two concurrent 512-clip forwards (two module
instances with the same weights, each with its own workspace), same total work.

    python tools/two_stream_probe.py [--batch 1024] [--steps 10]
"""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--parts", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    c = pkg.synth.CONFIGS["cfg2"]
    models = []
    sd = None
    for _ in range(args.parts):
        m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("cfg2")))
        if sd is None:
            sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234)
        m.load_state_dict(sd)
        models.append(m.train(False).to(dev))
    B = args.batch
    batch = {k: v.to(dev) for k, v in pkg.synth.make_batch(B, c["T"], c["N"], seed=1000).items()}
    h = B // args.parts
    parts = [{k: v[i * h:(i + 1) * h].contiguous() for k, v in batch.items()} for i in range(args.parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.parts)]

    def one():
        with torch.no_grad():
            return models[0](batch)["stlt"]

    def split():
        cur = torch.cuda.current_stream(dev)
        outs = []
        with torch.no_grad():
            for m, p, s in zip(models, parts, streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs.append(m(p)["stlt"])
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(outs)

    ref = one()
    got = split()
    print("max abs diff one vs split:", (ref - got).abs().max().item())
    for name, fn in (("one stream, %d clips" % B, one), ("%d streams x %d clips" % (args.parts, h), split), ("one stream, %d clips" % B, one),
                     ("%d streams x %d clips" % (args.parts, h), split)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(f"{name:28s} {dt * 1e3:8.2f} ms/step  {B / dt:9.1f} clips/s", flush=True)


if __name__ == "__main__":
    main()
