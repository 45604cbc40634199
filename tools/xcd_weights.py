#!/usr/bin/env python3
"""Relative speed of the eight XCDs under the f32-MFMA GEMM (GPU box only): per-workgroup tiles / elapsed time of a few
whole-tile launches (the in-kernel 100 MHz stamps), averaged per XCD (blockIdx & 7) and normalised to mean 1.
Prints the comma-separated list STLT_GEMM_XCD_W takes.

    STLT_GEMM_XCD_W=$(python tools/xcd_weights.py) python bench.py ...
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("STLT_GEMM_XCD_W", None)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
lib = pkg._lib.load()
M, N, K = 229376, 768, 768
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
for _ in range(20):  # warm: clocks settle under load
    pkg.ops.linear(x, w, b, out=y)
buf = torch.zeros(4 * 4096 + 8192, dtype=torch.int64, device="cuda")
G = 256
rates = torch.zeros(8, dtype=torch.float64)
n = 0
for _ in range(5):
    buf.zero_()
    lib.stlt_debug_set_buffer(buf.data_ptr())
    pkg.ops.linear(x, w, b, out=y)
    torch.cuda.synchronize()
    lib.stlt_debug_set_buffer(None)
    t = buf[: 4 * G].view(G, 4).cpu()
    dur = (t[:, 1] - t[:, 0]).double()
    r = t[:, 3].double() / dur
    for xcd in range(8):
        rates[xcd] += r[torch.arange(G) % 8 == xcd].mean()
    n += 1
    for _ in range(3):
        pkg.ops.linear(x, w, b, out=y)
rates /= n
rates /= rates.mean()
print(",".join(f"{v:.4f}" for v in rates.tolist()))
