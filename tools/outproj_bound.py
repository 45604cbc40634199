#!/usr/bin/env python3
"""Row N1(c), measured bound: what could an out-projection INSIDE the fused MHSA kernel save at most?  (GPU box only)

The fused kernel would keep the attention output (ctx) on chip instead of writing it, and the out-projection would not read it back and
would not be a launch of its own.  Measured here, per layer, at the temporal tower's shapes:
  (a) the fused kernel as shipped vs a timing build that computes ctx but does not store it (build/variants/libstlt_hip_mnoctx.so:
      -DSTLT_MHSA_ABLATE=1)                                                  -> the cost of the ctx write;
  (b) the out-projection + residual product (stlt_linear_fwd) on a ctx buffer that the previous launch just wrote (Infinity-Cache-hot) vs on
      one that 512 MB of other traffic has pushed out                         -> the cost of the ctx read;
  (c) back-to-back fused kernel + out-projection vs the sum of the two alone  -> the launch boundary.
Everything else of the out-projection (its 2 S L d^2 FLOPs) would still have to be computed inside the fused kernel, on 64-row items
(12 heads of a row group per workgroup: docs/LABNOTES_rounds1-5.md)."""
import ctypes as C
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build", "variants")


def load(tag):
    lib = C.CDLL(os.path.join(VDIR, f"libstlt_hip_{tag}.so"))
    vp = C.c_void_p
    lib.stlt_mhsa_fused_fwd_ex.restype = C.c_int
    lib.stlt_mhsa_fused_fwd_ex.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, vp, vp, vp]
    lib.stlt_linear_fwd.restype = C.c_int
    lib.stlt_linear_fwd.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, vp]
    return lib


def timed(fn, iters=20, rounds=5, between=None):
    for _ in range(3):
        fn()
    out = []
    for _ in range(rounds):
        tot = 0.0
        for _ in range(iters):
            if between is not None:
                between()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        out.append(tot / iters * 1e3)
    return statistics.median(out)


def main():
    tree, noctx = load("tree"), load("mnoctx")
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    H, d = 12, 768
    stream = torch.cuda.current_stream().cuda_stream
    junk = torch.empty(128 << 20, device=dev)  # 512 MB: twice the Infinity Cache
    for S, L in ((1024, 32), (256, 32), (1024, 17)):
        M = S * L
        x = torch.rand(M, d, device=dev, generator=g) * 2 - 1
        w = (torch.rand(3 * d, d, device=dev, generator=g) * 2 - 1) / d ** 0.5
        b = torch.rand(3 * d, device=dev, generator=g) - 0.5
        wo = (torch.rand(d, d, device=dev, generator=g) * 2 - 1) / d ** 0.5
        bo = torch.rand(d, device=dev, generator=g) - 0.5
        kpm = torch.zeros(S, L, device=dev, dtype=torch.uint8)
        ctx, y = torch.empty(M, d, device=dev), torch.empty(M, d, device=dev)

        def fused(lib):
            rc = lib.stlt_mhsa_fused_fwd_ex(x.data_ptr(), w.data_ptr(), b.data_ptr(), kpm.data_ptr(), 1, S, L, H, d, 0.0, 0, 0, ctx.data_ptr(), None, stream)
            assert rc == 0

        def outproj():
            rc = tree.stlt_linear_fwd(ctx.data_ptr(), d, wo.data_ptr(), bo.data_ptr(), y.data_ptr(), d, M, d, d, 0, stream)
            assert rc == 0

        t_fused, t_noctx = timed(lambda: fused(tree)), timed(lambda: fused(noctx))
        fused(tree)
        t_out_hot = timed(outproj, between=lambda: fused(tree))   # ctx just written by the producer
        t_out_cold = timed(outproj, between=lambda: junk.zero_())  # ctx (and the weights) evicted
        t_pair = timed(lambda: (fused(tree), outproj()))
        rec = {"sequences": S, "tokens": L, "fused_us": round(t_fused, 1), "fused_without_ctx_store_us": round(t_noctx, 1),
               "ctx_write_cost_us": round(t_fused - t_noctx, 1), "out_proj_ctx_hot_us": round(t_out_hot, 1), "out_proj_ctx_cold_us": round(t_out_cold, 1),
               "ctx_read_cost_us_at_most": round(t_out_cold - t_out_hot, 1), "fused_then_out_proj_us": round(t_pair, 1),
               "launch_boundary_us": round(t_pair - t_fused - t_out_hot, 1)}
        rec["saving_bound_us"] = round(max(rec["ctx_write_cost_us"], 0) + max(rec["launch_boundary_us"], 0), 1)
        rec["saving_bound_fraction_of_the_pair"] = round(rec["saving_bound_us"] / t_pair, 4)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
