#!/usr/bin/env python3
"""Per-shape timing of the three products of nn.Linear in a training step (GPU box only): y = x·Wᵀ + b (forward),
dx = dy·W and dW = dyᵀ·x, at the token counts of `bench.py --mode train` (cfg2, 64 clips: 14336 spatial / 2048 temporal
tokens), stream-K scratch lent as inside the model, fix-up launches included in the time.

    python tools/bench_gemm_train.py [--batch 64] [--iters 20]
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--T", type=int, default=32)
    ap.add_argument("--N", type=int, default=7)
    args = ap.parse_args()
    d, B, T, N = args.d, args.batch, args.T, args.N
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    tot_f = tot_t = 0.0
    with pkg.ops.gemm_scratch(dev):
        for tower, M, layers in (("sp", B * T * N, 4), ("tp", B * T, 8)):
            for name, n_out, k_in, act in (("qkv", 3 * d, d, 0), ("out", d, d, 0), ("ffn1", 4 * d, d, 1), ("ffn2", d, 4 * d, 0)):
                x = torch.rand(M, k_in, device=dev, generator=g) * 2 - 1
                w = (torch.rand(n_out, k_in, device=dev, generator=g) * 2 - 1) / k_in ** 0.5
                b = torch.rand(n_out, device=dev, generator=g)
                dy = torch.rand(M, n_out, device=dev, generator=g) * 2 - 1
                y = torch.empty(M, n_out, device=dev)
                fl = 2.0 * M * n_out * k_in
                for kind, fn in (("fwd", lambda: pkg.ops.linear(x, w, b, act=act, out=y)),
                                 ("dx", lambda: pkg.ops.gemm(dy, w, trans_b=True)),
                                 ("dw", lambda: pkg.ops.gemm(dy, x, trans_a=True, trans_b=True, k=M // 32 * 32))):
                    ms = timed(fn, args.iters)
                    tot_f += fl * layers
                    tot_t += ms * layers
                    print(f"{tower} {name:5s} {kind:3s} M={M:6d} out={n_out:5d} in={k_in:5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s  ({fl/ms/1e9/157.3:.3f})", flush=True)
        # the step itself does not run the four dW products of a layer one by one: they go out as ONE grouped stream-K launch
        for tower, M in (("sp", B * T * N), ("tp", B * T)):
            Mp = M // 32 * 32
            items, fl = [], 0.0
            for n_out, k_in in ((3 * d, d), (d, d), (4 * d, d), (d, 4 * d)):
                dy = torch.rand(Mp, n_out, device=dev, generator=g) * 2 - 1
                x = torch.rand(Mp, k_in, device=dev, generator=g) * 2 - 1
                items.append((dy, x, torch.zeros(n_out, k_in, device=dev)))
                fl += 2.0 * Mp * n_out * k_in
            ms = timed(lambda: pkg.ops.weight_grad_group(items), args.iters)
            print(f"{tower} 4 x dw, grouped launch  M={M:6d}                {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s  ({fl/ms/1e9/157.3:.3f})", flush=True)
    print(f"step total ({4}+{8} layers, products one at a time) {tot_t:.3f} ms  {tot_f/tot_t/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
