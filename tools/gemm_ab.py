#!/usr/bin/env python3
"""A/B of GEMM build variants in ONE process, interleaved rounds (GPU box only; build the variants on the CPU side).

    python tools/gemm_ab.py --build s0:-DSTLT_GEMM_STAGGER=0 s1:-DSTLT_GEMM_STAGGER=1     # CPU: build/variants/*.so
    python tools/gemm_ab.py --run s0 s1 [--batch 1024] [--rounds 7] [--iters 5]           # GPU: median TFLOP/s per shape

Every variant's output is compared with the first variant's (max abs difference), so a schedule change that breaks a
tile shows up here before the test-suite run.
"""
import argparse
import ctypes as C
import importlib
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build", "variants")


def build(specs):
    b = importlib.import_module("revisiting-spatial-temporal-layouts_amd.build")
    for spec in specs:
        tag, _, flags = spec.partition(":")
        fl = [f for f in flags.split(",") if f]
        print(tag, fl, b.variant(tag, {"gemm.hip": fl}), flush=True)


def run(tags, batch, rounds, iters, d, T, N, only):
    import torch
    libs = {}
    for t in tags:
        lib = C.CDLL(os.path.join(VDIR, f"libstlt_hip_{t}.so"))
        lib.stlt_linear_fwd.restype = C.c_int
        lib.stlt_linear_fwd.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_int64, C.c_int, C.c_void_p]
        lib.stlt_gemm_set_scratch.argtypes = [C.c_void_p, C.c_size_t]
        lib.stlt_debug_set_buffer.argtypes = [C.c_void_p]
        lib.stlt_gemm_scratch_bytes.restype = C.c_size_t
        libs[t] = lib
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = []
    M = batch * T * N
    shapes += [("sp qkv", M, 3 * d, d, 0), ("sp out", M, d, d, 0), ("sp ffn1", M, 4 * d, d, 1), ("sp ffn2", M, d, 4 * d, 0)]
    M = batch * T
    shapes += [("tp qkv", M, 3 * d, d, 0), ("tp out", M, d, d, 0), ("tp ffn1", M, 4 * d, d, 1), ("tp ffn2", M, d, 4 * d, 0)]
    if only:
        shapes = [s for s in shapes if any(o in s[0] for o in only)]
    scratch = torch.empty(libs[tags[0]].stlt_gemm_scratch_bytes(), dtype=torch.uint8, device=dev)
    for lib in libs.values():
        lib.stlt_gemm_set_scratch(scratch.data_ptr(), scratch.numel())
    stream = torch.cuda.current_stream().cuda_stream
    tot = {t: [0.0, 0.0] for t in tags}
    # in-kernel clock: with a debug buffer set, thread 0 of every workgroup leaves s_memrealtime (100 MHz) and s_memtime
    # (shader clock) at its start and end (4 stores per workgroup); the last launch of each timed burst is read back
    G, WAVES = 256, 8
    dbg = torch.zeros(4 * 4096 + 8192, dtype=torch.int64, device=dev)
    for lib in libs.values():
        lib.stlt_debug_set_buffer(dbg.data_ptr())

    def clock_ghz():
        t = dbg[: 4 * G].view(G, 4).cpu()
        ck = dbg[4 * G + G * WAVES * 6 + 1024: 4 * G + G * WAVES * 6 + 1024 + 2 * G].view(G, 2).cpu().double()
        return float(((ck[:, 1] - ck[:, 0]) / ((t[:, 1] - t[:, 0]).double() / 100.0)).median() / 1e3)

    for name, M, Nn, K, act in shapes:
        x = torch.rand(M, K, device=dev, generator=g) * 2 - 1
        w = (torch.rand(Nn, K, device=dev, generator=g) * 2 - 1) / K ** 0.5
        b = torch.rand(Nn, device=dev, generator=g)
        ys = {t: torch.empty(M, Nn, device=dev) for t in tags}

        def call(t):
            rc = libs[t].stlt_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), ys[t].data_ptr(), Nn, M, Nn, K, act, stream)
            assert rc == 0, (t, rc)

        for t in tags:
            call(t); call(t)
        torch.cuda.synchronize()
        diffs = {t: float((ys[t] - ys[tags[0]]).abs().max()) for t in tags}
        times = {t: [] for t in tags}
        clocks = {t: [] for t in tags}
        for _ in range(rounds):
            for t in tags:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    call(t)
                e1.record()
                torch.cuda.synchronize()
                times[t].append(e0.elapsed_time(e1) / iters)
                clocks[t].append(clock_ghz())
        fl = 2.0 * M * Nn * K
        line = f"{name:8s} M={M:7d} N={Nn:5d} K={K:5d}"
        for t in tags:
            med, mn = statistics.median(times[t]), min(times[t])
            tot[t][0] += fl; tot[t][1] += med
            ck = statistics.median(clocks[t])
            line += f" | {t}: {fl/med/1e9:6.1f} TF @{ck:.3f} GHz = {fl/med/1e9/(ck/2.4*157.3):.3f} d={diffs[t]:.0e}"
        print(line, flush=True)
    print("total: " + " | ".join(f"{t}: {tot[t][0]/tot[t][1]/1e9:6.1f} TF ({tot[t][1]:.3f} ms)" for t in tags))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", nargs="*")
    ap.add_argument("--run", nargs="*")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--T", type=int, default=32)
    ap.add_argument("--N", type=int, default=7)
    ap.add_argument("--only", nargs="*")
    a = ap.parse_args()
    if a.build:
        build(a.build)
    if a.run:
        run(a.run, a.batch, a.rounds, a.iters, a.d, a.T, a.N, a.only)
