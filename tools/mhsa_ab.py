#!/usr/bin/env python3
"""A/B of library builds on the fused in-projection + attention kernel (csrc/mhsa.hip) in ONE process, interleaved rounds (GPU box only).

    python tools/mhsa_ab.py <tagA> <tagB> ...      # build/variants/libstlt_hip_<tag>.so; "tree" = the in-tree library

Shapes: the temporal tower's launch (causal, T frames x clips) and the spatial tower's (key padding only, N slots x frames) at
cfg2 / the reference's layouts; us per launch (median of the rounds) and the max abs difference from the first build's output."""
import ctypes as C
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "build", "variants")
PKG = os.path.join(ROOT, "revisiting-spatial-temporal-layouts_amd")


def load(tag):
    lib = C.CDLL(os.path.join(PKG, "libstlt_hip.so") if tag == "tree" else os.path.join(VDIR, f"libstlt_hip_{tag}.so"))
    vp = C.c_void_p
    lib.stlt_mhsa_fused_fwd_ex.restype = C.c_int
    lib.stlt_mhsa_fused_fwd_ex.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, vp, vp, vp]
    return lib


def main(tags, rounds=7, iters=10):
    libs = {t: load(t) for t in tags}
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    H, d = 12, 768
    stream = torch.cuda.current_stream().cuda_stream
    for name, causal, S, L in (("temporal T=32 x 1024", 1, 1024, 32), ("temporal T=17 x 1024", 1, 1024, 17), ("temporal T=32 x 256", 1, 256, 32),
                               ("spatial N=7 x 32768", 0, 32768, 7), ("spatial N=5 x 17408", 0, 17408, 5), ("spatial N=8 x 33792", 0, 33792, 8)):
        x = torch.rand(S * L, d, device=dev, generator=g) * 2 - 1
        w = (torch.rand(3 * d, d, device=dev, generator=g) * 2 - 1) / d ** 0.5
        b = torch.rand(3 * d, device=dev, generator=g) - 0.5
        kpm = (torch.rand(S, L, device=dev, generator=g) < 0.2).to(torch.uint8)
        kpm[:, 0] = 0
        outs = {t: torch.empty(S * L, d, device=dev) for t in tags}

        def call(t):
            rc = libs[t].stlt_mhsa_fused_fwd_ex(x.data_ptr(), w.data_ptr(), b.data_ptr(), kpm.data_ptr(), causal, S, L, H, d, 0.0, 0, 0, outs[t].data_ptr(), None, stream)
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
        print(json.dumps({"shape": name, "us": {t: round(statistics.median(v), 1) for t, v in us.items()},
                          "max_abs_diff_vs_first": {t: float((outs[t] - outs[tags[0]]).abs().max()) for t in tags}}), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["tree"])
