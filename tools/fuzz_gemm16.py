#!/usr/bin/env python3
"""Random-shape check of the small-tile product kernel (csrc/gemm16*.hip) on every tile shape (GPU box only): forward with bias / GELU / ReLU /
residual and the input gradient (weight read as it lies), ragged M / N, K = 64 ... 1248, against an fp64 product and for bit-identity of
two launches.  One line per failure, a summary line at the end.

    python tools/fuzz_gemm16.py [--n 600] [--seed 0]
"""
import argparse, importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=600); ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(a.seed)
    dev = "cuda"
    tiles = pkg.ops.SMALL_TILES
    bad, worst = 0, 0.0
    for i in range(a.n):
        tr, tc = tiles[i % len(tiles)]
        M = int(torch.randint(1, 3000, (1,), generator=g)) if i % 5 else int(torch.randint(1, 40000, (1,), generator=g))
        N = 4 * int(torch.randint(1, 300, (1,), generator=g))
        K = 32 * int(torch.randint(2, 40, (1,), generator=g))
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev)
        ref = x.double() @ w.double().t()
        tol = 2.0 ** -23 * math.sqrt(K) * max(ref.abs().max().item(), 1.0) * 4
        mode = i % 4
        if mode == 0:
            got = pkg.ops.linear_small(x, w, b, tc, tile_rows=tr); again = pkg.ops.linear_small(x, w, b, tc, tile_rows=tr); want = ref + b.double()
        elif mode == 1:
            got = pkg.ops.linear_small(x, w, b, tc, act=1, tile_rows=tr); again = pkg.ops.linear_small(x, w, b, tc, act=1, tile_rows=tr)
            want = torch.nn.functional.gelu(ref + b.double())
        elif mode == 2:
            got = pkg.ops.linear_small(x, w, None, tc, residual=r, tile_rows=tr); again = pkg.ops.linear_small(x, w, None, tc, residual=r, tile_rows=tr)
            want = ref + r.double()
        else:  # input gradient: dx (M, N) = dy (M, K) · w2 (K, N)
            w2 = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(dev)
            got = pkg.ops.input_grad_small(x, w2, tc, tile_rows=tr); again = pkg.ops.input_grad_small(x, w2, tc, tile_rows=tr)
            want = x.double() @ w2.double()
            tol = 2.0 ** -23 * math.sqrt(K) * max(want.abs().max().item(), 1.0) * 4
        err = (got.double() - want).abs().max().item()
        worst = max(worst, err / tol)
        if not (torch.isfinite(got).all().item() and torch.equal(got, again) and err <= tol):
            bad += 1
            print(f"FAIL tile {tr}x{tc} mode {mode} M={M} N={N} K={K}: err {err:.3e} tol {tol:.3e} identical {torch.equal(got, again)}", flush=True)
    print(f"{a.n} random shapes over {len(tiles)} tiles: {bad} failures; worst error / bound = {worst:.2f}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
