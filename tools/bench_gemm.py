#!/usr/bin/env python3
"""Per-shape timing of stlt_linear_fwd on the GEMM shapes of the STLT forward (GPU box only).

    python tools/bench_gemm.py [--batch 256] [--iters 20] [--shapes spatial|temporal|all]
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--T", type=int, default=32)
    ap.add_argument("--N", type=int, default=7)
    args = ap.parse_args()
    d, B, T, N = args.d, args.batch, args.T, args.N
    shapes = []
    if args.shapes in ("all", "spatial"):
        M = B * T * N
        shapes += [("sp qkv", M, 3 * d, d, 0), ("sp out", M, d, d, 0), ("sp ffn1", M, 4 * d, d, 1), ("sp ffn2", M, d, 4 * d, 0)]
    if args.shapes in ("all", "temporal"):
        M = B * T
        shapes += [("tp qkv", M, 3 * d, d, 0), ("tp out", M, d, d, 0), ("tp ffn1", M, 4 * d, d, 1), ("tp ffn2", M, d, 4 * d, 0)]
    dev = "cuda"
    # lend the stream-K scratch, as the whole-path entry points do from their workspace: under-filled launches then run
    # the way they run inside the model
    lib = pkg._lib.load()
    scratch = torch.empty(int(lib.stlt_gemm_scratch_bytes()), dtype=torch.uint8, device=dev)
    pkg._lib.check(lib.stlt_gemm_set_scratch(scratch.data_ptr(), scratch.numel()), "stlt_gemm_set_scratch")
    g = torch.Generator(device=dev).manual_seed(0)
    tot_f = tot_t = 0.0
    for name, M, Nn, K, act in shapes:
        x = torch.rand(M, K, device=dev, generator=g) * 2 - 1
        w = (torch.rand(Nn, K, device=dev, generator=g) * 2 - 1) / K ** 0.5
        b = torch.rand(Nn, device=dev, generator=g)
        y = torch.empty(M, Nn, device=dev)
        for _ in range(3):
            pkg.ops.linear(x, w, b, act=act, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            pkg.ops.linear(x, w, b, act=act, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        fl = 2.0 * M * Nn * K
        tot_f += fl
        tot_t += ms
        print(f"{name:8s} M={M:7d} N={Nn:5d} K={K:5d} act={act}  {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
    print(f"total {tot_t:.3f} ms  {tot_f/tot_t/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
