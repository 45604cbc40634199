#!/usr/bin/env python3
"""Stress of the stream-K tile completion (GPU box only): the same under-filled products, new random operands every
round, compared with a float64 reference and with a second run (bitwise).  A visibility race between the workgroups of a
tile would show up as a wrong or unstable tile.

    python tools/sk_stress.py [--rounds 30]
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
    ap.add_argument("--rounds", type=int, default=30)
    args = ap.parse_args()
    dev = "cuda"
    shapes = [(2048, 768, 768), (2048, 768, 3072), (2048, 2304, 768), (14336, 768, 768), (14336, 3072, 768), (64, 768, 768), (1000, 174, 256)]
    g = torch.Generator(device=dev).manual_seed(0)
    bad = 0
    with pkg.ops.gemm_scratch(dev):
        for r in range(args.rounds):
            for M, N, K in shapes:
                x = torch.rand(M, K, device=dev, generator=g) * 2 - 1
                w = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5
                b = torch.rand(N, device=dev, generator=g)
                y1 = pkg.ops.linear(x, w, b)
                y2 = pkg.ops.linear(x, w, b)
                if N % 4:  # the contraction-major layout wants 16-byte aligned rows
                    continue_dw = True
                    dw1 = dw2 = None
                else:
                    continue_dw = False
                    dw1 = pkg.ops.gemm(y1, x, trans_a=True, trans_b=True, k=M // 32 * 32)
                    dw2 = pkg.ops.gemm(y1, x, trans_a=True, trans_b=True, k=M // 32 * 32)
                ref = x.double() @ w.double().t() + b.double()
                err = (y1.double() - ref).abs().max().item()
                errw, stable_w = 0.0, True
                if not continue_dw:
                    refw = y1[: M // 32 * 32].double().t() @ x[: M // 32 * 32].double()
                    errw = ((dw1.double() - refw).abs().max() / refw.abs().max()).item()
                    stable_w = torch.equal(dw1, dw2)
                ok = err <= 2e-5 and errw <= 2e-5 and torch.equal(y1, y2) and stable_w
                if not ok:
                    bad += 1
                    print(f"round {r} shape {(M, N, K)}: err {err:.3e} errw {errw:.3e} stable {torch.equal(y1, y2)} {stable_w}", flush=True)
    print(f"{args.rounds} rounds x {len(shapes)} shapes x 4 launches: {bad} bad")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
