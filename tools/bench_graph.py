#!/usr/bin/env python3
"""Eager call against hipGraph replay of the STLT forward at small batches (GPU box only): where the forward is bound by its ~100 launches'
host time, a replay of the captured sequence removes it.  One JSON line per batch.

    python tools/bench_graph.py [--config cfg2] [--batches 1 8 64] [--iters 200]
"""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2"); ap.add_argument("--batches", type=int, nargs="+", default=[1, 8, 64]); ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    c = pkg.synth.CONFIGS[a.config]
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(a.config)))
    m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234))
    m.train(False).to("cuda")
    for B in a.batches:
        batch = {k: v.to("cuda") for k, v in pkg.synth.make_batch(B, c["T"], c["N"], seed=B).items()}
        with torch.no_grad():
            for _ in range(5):
                ref = m(batch)["stlt"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                m(batch)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / a.iters
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m(batch)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = m(batch)["stlt"]
            g.replay(); torch.cuda.synchronize()
            same = bool(torch.equal(out, ref))
            t0 = time.perf_counter()
            for _ in range(a.iters):
                g.replay()
            torch.cuda.synchronize()
            replay = (time.perf_counter() - t0) / a.iters
        print(json.dumps({"config": a.config, "clips": B, "eager_ms": round(eager * 1e3, 4), "graph_replay_ms": round(replay * 1e3, 4),
                          "eager_clips_per_s": round(B / eager, 1), "replay_clips_per_s": round(B / replay, 1), "bit_identical": same}), flush=True)


if __name__ == "__main__":
    main()
