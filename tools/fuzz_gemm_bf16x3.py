#!/usr/bin/env python3
"""Random-shape check of the split-bf16 kernel against the f32-MFMA kernel and an fp64 product (GPU box only).
Run with STLT_X3_MIN_FILL=0 so that every launch — however small or ragged — goes to the split kernel:

    STLT_X3_MIN_FILL=0 python tools/fuzz_gemm_bf16x3.py [--n 300] [--seed 0]
"""
import argparse, importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=300); ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(a.seed)
    dev = "cuda"
    worst, taken = 0.0, 0
    with pkg.ops.gemm_scratch(dev):
        for i in range(a.n):
            M = int(torch.randint(1, 5000, (1,), generator=g)) if i % 3 else int(torch.randint(1, 70000, (1,), generator=g))
            N = int(torch.randint(1, 1200, (1,), generator=g))
            K = 32 * int(torch.randint(2, 40, (1,), generator=g))
            ldx = K + 4 * int(torch.randint(0, 9, (1,), generator=g)) if i % 4 == 0 else K
            act = int(torch.randint(0, 3, (1,), generator=g))
            with_bias = bool(torch.randint(0, 2, (1,), generator=g))
            xs = torch.randn(M, ldx, generator=g).to(dev)
            w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
            b = torch.randn(N, generator=g).to(dev) if with_bias else None
            kw = dict(rows=M, ldx=ldx) if ldx != K else {}
            x = xs if ldx == K else xs
            pkg.ops.set_gemm_split_bf16(0); y0 = pkg.ops.linear(x, w, b, act=act, **kw)
            pkg.ops.set_gemm_split_bf16(6); y6 = pkg.ops.linear(x, w, b, act=act, **kw)
            ref = xs[:, :K].double() @ w.double().t()
            if with_bias: ref = ref + b.double()
            if act == 1: ref = torch.nn.functional.gelu(ref)
            if act == 2: ref = torch.relu(ref)
            e0, e6 = (y0.double() - ref).abs().max().item(), (y6.double() - ref).abs().max().item()
            taken += int(not torch.equal(y0, y6))
            # bar: twice the f32 kernel's error, or eps_f32 sqrt(K) of the largest output — what one sequential f32 accumulation
            # over K may lose (the f32 kernel's small launches run as stream-K, K cut into ranges summed pairwise, and land 2-3x
            # under that; on long whole-tile launches the two kernels' errors are equal: tests/test_gemm_bf16x3_gpu.py)
            ok = torch.isfinite(y6).all().item() and e6 <= max(2.0 * e0, 2.0 ** -23 * math.sqrt(K) * max(ref.abs().max().item(), 1.0))
            worst = max(worst, e6 / max(e0, 1e-9))
            if not ok:
                print(f"FAIL M={M} N={N} K={K} ldx={ldx} act={act} bias={with_bias}: err f32 {e0:.3e} split {e6:.3e}", flush=True)
                sys.exit(1)
    pkg.ops.set_gemm_split_bf16(0)
    print(f"{a.n} shapes ok ({taken} of them differed from the f32 kernel's bits, i.e. ran on the split kernel); worst error ratio split / f32 = {worst:.2f}")


if __name__ == "__main__":
    main()
