#!/usr/bin/env python3
"""Fit the small-tile kernel's launch-time estimate (csrc/gemm16.hip: est16_us) to a tools/bench_gemm16.py run.

    python tools/fit_gemm16.py gpurun_out/.../gemm16_shapes.jsonl          # prints the C table for csrc/gemm16.hip + residuals

Model per tile (rows x columns) and build (forward / input gradient):  us = a + rounds * (nk * s + e)
with rounds = ceil(tiles / 256 CUs), nk = K / 32: a = launch boundary + pipeline fill, s = one k-step, e = tile epilogue / restart.
Least squares over every measured shape of that tile; the table is what the routing compares with the large-tile estimate."""
import json, math, sys
import numpy as np

TILES = tuple((128, c) for c in (48, 64, 96, 128, 144, 192)) + tuple((64, c) for c in (64, 96, 128, 160, 192, 256)) + tuple((32, c) for c in (128, 192, 256))
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
out = {}
for dx in (False, True):
    for tr, tc in TILES:
        A, y = [], []
        for r in rows:
            if r["shape"].endswith("_dx") != dx:
                continue
            tiles = math.ceil(r["M"] / tr) * math.ceil(r["N"] / tc)
            rounds = math.ceil(tiles / 256)
            nk = r["K"] // 32
            A.append([1.0, rounds * nk, rounds]); y.append(r[f"t{tr}x{tc}_us"])
        A, y = np.array(A), np.array(y)
        w = 1.0 / y  # relative error
        coef, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
        pred = A @ coef
        rel = np.abs(pred - y) / y
        out[(dx, tr, tc)] = (coef, rel.max(), rel.mean())
for dx in (False, True):
    print("// %s: {rows, cols, a (us), s (us per k-step), e (us per tile)}   max / mean relative error of the fit" % ("input gradient (WKN)" if dx else "forward"))
    for tr, tc in TILES:
        c, mx, mean = out[(dx, tr, tc)]
        ideal = 2.0 * tr * tc * 32 / 0.6144e6  # one k-step at the matrix pipe's 0.6144 TFLOP/s per CU
        print(f"  {{{tr}, {tc}, {c[0]:.2f}, {c[1]:.4f}, {c[2]:.2f}}},  // {mx:.3f} / {mean:.3f}; k-step at the MFMA rate {ideal:.4f} us -> {ideal / c[1]:.2f}")
