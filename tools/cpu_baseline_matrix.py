#!/usr/bin/env python3
"""CPU baseline matrix of BASELINE.md §3, timed on the GPU box's host cores: the oracle (oracle/stlt_oracle.py, the
build's CPU restatement of the reference forward, kind "port") on cfg1 (B=8), cfg2 (B=8, B=64) and cfg4 (B=4), eval mode,
no_grad, with torch.set_num_threads(n) for n = 1, the best of a few probed counts, and all host cores.

    python tools/cpu_baseline_matrix.py [--out gpurun_out/round2_cpu_baseline_matrix.json] [--budget 25]

One warm-up + up to three timed forwards per cell (fewer when one forward exceeds the per-cell budget in seconds).
This is a reported baseline, not a target; bench.py's own `cpu_baseline` object is one cell of this table (cfg2, 32 clips,
best thread count) so that the default bench run stays short.
"""
import argparse
import importlib
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "round2_cpu_baseline_matrix.json"))
    ap.add_argument("--budget", type=float, default=25.0, help="seconds of timed forwards per cell (at least one forward)")
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
    from oracle import stlt_oracle as O
    cores = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    cells = []
    for name, B in (("cfg1", 8), ("cfg2", 8), ("cfg2", 64), ("cfg4", 4)):
        c = pkg.synth.CONFIGS[name]
        model = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
        sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
        batch = pkg.synth.make_batch(B, c["T"], c["N"], dataset=c["dataset"], seed=1000)
        H = c["num_attention_heads"]

        def timed(n_threads, budget):
            torch.set_num_threads(n_threads)
            with torch.no_grad():
                O.stlt_forward(sd, batch, H)  # warm-up
                n_it, t0 = 0, time.perf_counter()
                while n_it < 1 or (n_it < 3 and time.perf_counter() - t0 < budget):
                    O.stlt_forward(sd, batch, H)
                    n_it += 1
                dt = (time.perf_counter() - t0) / n_it
            return dt, n_it

        probe = {}
        for t in sorted({t for t in (8, 16, 32, 64) if t <= cores}):
            probe[t] = timed(t, 0.0)[0]
        best = min(probe, key=probe.get)
        for label, n_threads in (("1 thread", 1), (f"best of {sorted(probe)} probed", best), ("all host cores", cores)):
            dt, n_it = timed(n_threads, args.budget)
            cells.append({"config": name, "batch": B, "threads": n_threads, "threads_label": label, "ms_per_forward": round(dt * 1e3, 2),
                          "clips_per_s": round(B / dt, 3), "timed_forwards": n_it})
            print(cells[-1], flush=True)
    torch.set_num_threads(default_threads)
    out = {"what": "oracle/stlt_oracle.py (CPU restatement of the reference forward; kind 'port'), eval mode, no_grad, fp32",
           "host": {"cpu": cpu_model(), "os_cpu_count": cores, "torch_default_threads": default_threads, "torch": torch.__version__},
           "cells": cells}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
