#!/usr/bin/env python3
"""What a launch costs when it does nothing: chains of dependent launches of the library's smallest kernels (exact-erf GELU over 4 ... 64 K
elements, residual + LayerNorm over 1 ... 64 rows), replayed from a hipGraph so that no host time is in the figure.  The per-launch time
of such a chain is the floor under every launch of a step — the BOUNDARY that tools/launch_bound.py grants 1.5 us for."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
lib = pkg._lib.load()
dev = torch.device("cuda", 0)
N_CHAIN, REPLAYS = 400, 20


def chain(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N_CHAIN):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPLAYS):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (REPLAYS * N_CHAIN)


out = {}
for n in (4, 1024, 65536):
    u, h = torch.randn(n, device=dev), torch.empty(n, device=dev)
    state = {"src": u, "dst": h}

    def gelu():
        pkg._lib.check(lib.stlt_gelu_fwd(state["src"].data_ptr(), state["dst"].data_ptr(), n, torch.cuda.current_stream().cuda_stream), "gelu")
        state["src"], state["dst"] = state["dst"], state["src"]  # every launch reads what the one before it wrote

    out[f"gelu_fwd n={n}"] = round(chain(gelu), 3)
for rows in (1, 64, 1088):
    x = torch.randn(rows, 768, device=dev)
    y = torch.empty_like(x)
    w, b = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    st = {"src": x, "dst": y}

    def ln():
        pkg._lib.check(lib.stlt_add_layernorm_fwd(st["src"].data_ptr(), 768, None, 0, w.data_ptr(), b.data_ptr(), 1e-5, rows, 768, st["dst"].data_ptr(), 768,
                                                  torch.cuda.current_stream().cuda_stream), "add_ln")
        st["src"], st["dst"] = st["dst"], st["src"]

    out[f"add_ln rows={rows}"] = round(chain(ln), 3)
# ... and in context: the out-projection of a temporal layer at 64 clips of 17 frames (1088 x 768 x 768 on the small tiles) alone, and followed by
# the residual + LayerNorm pass that reads its output — what the LayerNorm launch costs behind a product instead of behind itself
x = torch.randn(1088, 768, device=dev)
wq = torch.randn(768, 768, device=dev) / 28.0
bq = torch.zeros(768, device=dev)
y1, y2 = torch.empty_like(x), torch.empty_like(x)
w1, b1 = torch.ones(768, device=dev), torch.zeros(768, device=dev)


def product():
    pkg._lib.check(lib.stlt_linear_fwd(x.data_ptr(), 768, wq.data_ptr(), bq.data_ptr(), y1.data_ptr(), 768, 1088, 768, 768, 0, torch.cuda.current_stream().cuda_stream), "linear")


def product_then_ln():
    product()
    pkg._lib.check(lib.stlt_add_layernorm_fwd(y1.data_ptr(), 768, None, 0, w1.data_ptr(), b1.data_ptr(), 1e-5, 1088, 768, y2.data_ptr(), 768,
                                              torch.cuda.current_stream().cuda_stream), "add_ln")


out["linear 1088x768x768"] = round(chain(product), 3)
pair = chain(product_then_ln)
out["linear 1088x768x768 + add_ln rows=1088 (per pair)"] = round(pair, 3)
out["add_ln rows=1088 behind the product (pair - product)"] = round(pair - out["linear 1088x768x768"], 3)
print(json.dumps({"us_per_dependent_launch_in_a_graph_replay": out, "chain": N_CHAIN, "replays": REPLAYS}))
