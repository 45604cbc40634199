#!/usr/bin/env python3
"""Split-bf16 (six bf16 MFMA products per f32 product, csrc/gemm_bf16x3.hip) against the f32-MFMA kernel on the forward's
GEMM shapes: time per launch and error against an fp64 product (GPU box only).

    python tools/bench_gemm_bf16x3.py [--batch 1024] [--iters 20]
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
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--d", type=int, default=768)
    args = ap.parse_args()
    d, B = args.d, args.batch
    lib = pkg._lib.load()
    dev = "cuda"
    scratch = torch.empty(int(lib.stlt_gemm_scratch_bytes()), dtype=torch.uint8, device=dev)
    pkg._lib.check(lib.stlt_gemm_set_scratch(scratch.data_ptr(), scratch.numel()), "stlt_gemm_set_scratch")
    shapes = [("sp qkv", B * 224, 3 * d, d, 0, False), ("sp out", B * 224, d, d, 0, False), ("sp ffn1", B * 224, 4 * d, d, 1, False),
              ("sp ffn2", B * 224, d, 4 * d, 0, False), ("tp ffn1", B * 32, 4 * d, d, 1, False), ("tp ffn2", B * 32, d, 4 * d, 0, False),
              ("ragged", 12345, 777, 96, 0, False), ("small", 300, 130, 64, 1, False)]
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, N, K, act, add in shapes:
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        b = torch.randn(N, device=dev, generator=g)
        r = torch.randn(M, N, device=dev, generator=g) if add else None
        y = torch.empty(M, N, device=dev)

        def run():
            pkg.ops.linear(x, w, b, act=act, out=y)
        # fp64 on a row sample (the full product would take minutes)
        idx = torch.randperm(M, device=dev, generator=g)[: min(M, 2048)]
        ref = x[idx].double() @ w.double().t() + b.double()
        if add:
            ref = ref + r[idx].double()
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        res = {}
        for mode in (0, 6):
            pkg._lib.check(lib.stlt_set_gemm_split_bf16(mode), "split")
            y.zero_()
            ms = timed(run, args.iters)
            err = (y[idx].double() - ref).abs()
            res[mode] = (ms, err.max().item(), err.mean().item(), y.clone())
        pkg._lib.check(lib.stlt_set_gemm_split_bf16(0), "split")
        fl = 2.0 * M * N * K
        same = (res[0][3] == res[6][3]).float().mean().item()
        print(f"{name:10s} M={M:7d} N={N:5d} K={K:5d}  f32 {res[0][0]*1e3:8.1f} us {fl/res[0][0]/1e9:6.1f} TF/s err max {res[0][1]:.2e} mean {res[0][2]:.2e} | "
              f"bf16x3 {res[6][0]*1e3:8.1f} us {fl/res[6][0]/1e9:6.1f} TF/s err max {res[6][1]:.2e} mean {res[6][2]:.2e} | x{res[0][0]/res[6][0]:.2f} identical {same:.3f}", flush=True)


if __name__ == "__main__":
    main()
