#!/usr/bin/env python3
"""Small-tile product kernel (csrc/gemm16.hip) against the large-tile kernel (stream-K + fix-up where under-filled), per shape and
tile width (GPU box only).  Prints one row per shape: us per launch of stlt_linear_fwd's large-tile path (routing switched off) and
of every tile width, and the tile width the dispatch's launch-time estimate picks (0 = large tiles).

    python tools/bench_gemm16.py [--rows 2048 2112 4096] [--iters 50]
"""
import argparse, importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def timed(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


HEADER = """# Round 5: small-tile product kernel (csrc/gemm16*.hip: whole tiles of 128 / 64 / 32 rows) against the large-tile kernel (csrc/gemm.hip: 256 x 128
# tiles; stream-K + fix-up launch where under-filled).  MI355X, d = 768, stand-alone launches (tools/bench_gemm16.py --dx, 50 timed launches each,
# torch events around both launches of a stream-K product); us per launch.  RxC = tile rows x columns; choice = what the launch-time estimate
# picks (large = 256 x 128 tiles); TF/s of the large-tile launch, of the best measured tile and of the routed choice.
# *_dx rows: input gradient dX = dY·W with W (n_out = K, k_in = N) read as it lies (WKN build) against gemm.hip's NN kernel.
"""
TILES = tuple((128, c) for c in (48, 64, 96, 128, 144, 192)) + tuple((64, c) for c in (64, 96, 128, 160, 192, 256)) + tuple((32, c) for c in (128, 192, 256))


def tname(tr, tc):
    return f"t{tr}x{tc}_us"


def choice_name(tile):
    return "large" if tile == 0 else f"{(tile >> 16) or 128}x{tile & 0xffff}"


def table(path):
    """The committed text form of a run's JSON lines (profiles/round5_gemm16_shapes.txt): python tools/bench_gemm16.py --table <jsonl>"""
    rows = [json.loads(l) for l in open(path) if l.startswith("{")]
    if "--rechoose" in sys.argv:  # the routing of the library in the tree instead of the run's (pure host arithmetic: works without a GPU)
        lib = pkg._lib.load()
        for r in rows:
            dx = r["shape"].endswith("_dx")
            r["choice"] = choice_name(int(lib.stlt_input_grad_small_choice(r["M"], r["K"], r["N"]) if dx else lib.stlt_linear_small_choice(r["M"], r["N"], r["K"])))
    print(HEADER)
    print(f"{'M':>6} {'shape':>8} {'N':>5} {'K':>5} {'large':>7} " + " ".join(f"{'%dx%d' % t:>7}" for t in TILES) + "     best   choice  TF/s: large  best  routed")
    worst = 0.0
    for r in rows:
        t = {f"{tr}x{tc}": r[tname(tr, tc)] for tr, tc in TILES}
        best = min([r["large_us"]] + list(t.values()))
        routed = t[r["choice"]] if r["choice"] != "large" else r["large_us"]
        fl = 2.0 * r["M"] * r["N"] * r["K"] / 1e6
        print(f"{r['M']:>6} {r['shape']:>8} {r['N']:>5} {r['K']:>5} {r['large_us']:>7.1f} " + " ".join(f"{v:>7.1f}" for v in t.values())
              + f" {r['best']:>8} {r['choice']:>8} {fl / r['large_us']:>11.1f} {fl / best:>6.1f} {fl / routed:>6.1f}")
        worst = max(worst, routed / best)
    print(f"\nworst (time of the routed choice) / (best measured) over the {len(rows)} rows: {worst:.3f}")


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--table":
        return table(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[1088, 2048, 2112, 4096, 5440, 14336, 16896])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dx", action="store_true", help="also the input-gradient products (W read as it lies)")
    a = ap.parse_args()
    lib = pkg._lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    d = 768
    for M in a.rows:
        for name, N, K, act in (("qkv", 3 * d, d, 0), ("out", d, d, 0), ("ffn1", 4 * d, d, 1), ("ffn2", d, 4 * d, 0), ("kv", 2 * d, d, 0), ("in_dx", d, 3 * d, 0)):
            x = torch.rand(M, K, device=dev, generator=g) * 2 - 1
            w = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5
            b = torch.rand(N, device=dev, generator=g) - 0.5
            y = torch.empty(M, N, device=dev)
            row = {"M": M, "shape": name, "N": N, "K": K}
            pkg.ops.set_gemm_small_tiles(0)
            with pkg.ops.gemm_scratch(dev):
                row["large_us"] = round(timed(lambda: pkg.ops.linear(x, w, b, act=act, out=y), a.iters), 1)
            pkg.ops.set_gemm_small_tiles(-2)
            best = None
            for tr, tc in pkg.ops.SMALL_TILES:
                us = round(timed(lambda: pkg.ops.linear_small(x, w, b, tc, act=act, out=y, tile_rows=tr), a.iters), 1)
                row[tname(tr, tc)] = us
                if best is None or us < best[1]:
                    best = (f"{tr}x{tc}", us)
            row["best"] = best[0]
            row["choice"] = choice_name(int(lib.stlt_linear_small_choice(M, N, K)))
            row["tflops_large"] = round(2.0 * M * N * K / row["large_us"] / 1e6, 1)
            row["tflops_best"] = round(2.0 * M * N * K / best[1] / 1e6, 1)
            print(json.dumps(row), flush=True)
        if not a.dx:
            continue
        # the input-gradient products dX = dY·W of the same Linears (W (n_out, k_in) read as it lies): large NN kernel vs the WKN build
        for name, n_out, k_in in (("qkv_dx", 3 * d, d), ("out_dx", d, d), ("ffn1_dx", 4 * d, d), ("ffn2_dx", d, 4 * d), ("kv_dx", 2 * d, d)):
            dy = torch.rand(M, n_out, device=dev, generator=g) * 2 - 1
            w = (torch.rand(n_out, k_in, device=dev, generator=g) * 2 - 1) / n_out ** 0.5
            dx = torch.empty(M, k_in, device=dev)
            stream = torch.cuda.current_stream().cuda_stream
            row = {"M": M, "shape": name, "N": k_in, "K": n_out}
            def large():
                pkg._lib.check(lib.stlt_gemm(0, 1, dy.data_ptr(), n_out, w.data_ptr(), k_in, None, 0, dx.data_ptr(), k_in, 0, M, k_in, n_out, 1, stream), "stlt_gemm")
            with pkg.ops.gemm_scratch(dev):
                row["large_us"] = round(timed(large, a.iters), 1)
            best = None
            for tr, tc in pkg.ops.SMALL_TILES:
                us = round(timed(lambda: pkg.ops.input_grad_small(dy, w, tc, tile_rows=tr), a.iters), 1)
                row[tname(tr, tc)] = us
                if best is None or us < best[1]:
                    best = (f"{tr}x{tc}", us)
            row["best"] = best[0]
            row["choice"] = choice_name(int(lib.stlt_input_grad_small_choice(M, n_out, k_in)))
            row["tflops_large"] = round(2.0 * M * n_out * k_in / row["large_us"] / 1e6, 1)
            row["tflops_best"] = round(2.0 * M * n_out * k_in / best[1] / 1e6, 1)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
