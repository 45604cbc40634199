#!/usr/bin/env python3
"""Timing-only ablations / build variants of the small-tile product kernel (csrc/gemm16*.hip) in ONE process, interleaved rounds.

    python tools/gemm16_ab.py --build base: a1:-DSTLT_G16_ABLATE=1 a2:-DSTLT_G16_ABLATE=2 ...    # CPU: build/variants/*.so
    python tools/gemm16_ab.py --run base a1 a2 ... [--rounds 5] [--iters 20]                       # GPU: median us per (shape, tile)

STLT_G16_ABLATE bits (gemm16_kernel.h; results are WRONG in those builds, only the time means anything): 1 no steady-state DMA,
2 no steady-state fragment reads, 4 no steady-state barrier, 8 no epilogue stores.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build", "variants")
FILES = ("gemm16.hip", "gemm16_rb4.hip", "gemm16_rb2.hip")


def build(specs):
    b = importlib.import_module("revisiting-spatial-temporal-layouts_amd.build")
    for spec in specs:
        tag, _, flags = spec.partition(":")
        fl = [f for f in flags.split(",") if f]
        print(tag, fl, b.variant(tag, {f: fl for f in FILES}), flush=True)


CASES = [  # (M, N, K, act, tile rows, tile cols)
    (14336, 768, 768, 0, 128, 48), (14336, 768, 768, 0, 64, 96), (14336, 3072, 768, 1, 128, 192), (14336, 3072, 768, 1, 64, 256),
    (2048, 768, 768, 0, 128, 48), (2048, 3072, 768, 1, 128, 192), (2048, 768, 3072, 0, 128, 48),
    (1088, 768, 768, 0, 64, 64), (1088, 2304, 768, 0, 64, 160), (5440, 2304, 768, 0, 64, 256), (5440, 768, 3072, 0, 64, 256),
    (1088, 768, 3072, 0, 64, 64), (1088, 768, 768, 0, 32, 128), (2048, 2304, 768, 0, 128, 144), (14336, 768, 3072, 0, 128, 48), (2112, 768, 768, 0, 128, 64),
]


def run(tags, rounds, iters):
    import torch
    libs = {}
    for t in tags:
        lib = C.CDLL(os.path.join(VDIR, f"libstlt_hip_{t}.so"))
        lib.stlt_linear_small_fwd.restype = C.c_int
        lib.stlt_linear_small_fwd.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                              C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        libs[t] = lib
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    stream = torch.cuda.current_stream().cuda_stream
    for M, N, K, act, tr, tc in CASES:
        x = torch.rand(M, K, device=dev, generator=g) * 2 - 1
        w = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5
        b = torch.rand(N, device=dev, generator=g)
        y = torch.empty(M, N, device=dev)
        tile = tc if tr == 128 else (tc | (tr << 16))

        def call(t):
            rc = libs[t].stlt_linear_small_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), N, M, N, K, act, tile, stream)
            assert rc == 0, (t, rc)

        for t in tags:
            call(t); call(t)
        torch.cuda.synchronize()
        us = {t: [] for t in tags}
        for _ in range(rounds):
            for t in tags:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    call(t)
                e1.record()
                torch.cuda.synchronize()
                us[t].append(e0.elapsed_time(e1) / iters * 1e3)
        med = {t: round(statistics.median(v), 2) for t, v in us.items()}
        ideal = 2.0 * M * N * K / 157.3e6
        print(json.dumps({"M": M, "N": N, "K": K, "act": act, "tile": f"{tr}x{tc}", "ideal_us": round(ideal, 1), "us": med}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", nargs="+")
    ap.add_argument("--run", nargs="+")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    if a.build:
        build(a.build)
    if a.run:
        run(a.run, a.rounds, a.iters)
