#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the attention kernel from in-kernel s_memtime stamps (GPU box only)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
lib = pkg._lib.load()
NAMES = ["issue loads", "wait data", "QK^T", "mask+softmax", "P.V", "epilogue->LDS", "stores+drain"]
for (S, L, causal) in ((64, 32, True), (256, 32, True), (1024, 32, True), (32768, 7, False), (4096, 36, False)):
    H = 12; d = 64 * H
    qkv = torch.randn(S, L, 3 * d, device="cuda"); kpm = torch.zeros(S, L, dtype=torch.bool, device="cuda")
    P = 32 // L if L <= 16 else 1
    n_items = ((S + P - 1) // P) * ((P * L + 31) // 32) * H
    buf = torch.zeros(n_items * 8, dtype=torch.int64, device="cuda")
    for _ in range(3): pkg.ops.attn_core(qkv, kpm, causal, H)
    lib.stlt_debug_set_buffer(buf.data_ptr())
    pkg.ops.attn_core(qkv, kpm, causal, H); torch.cuda.synchronize()
    lib.stlt_debug_set_buffer(None)
    t = buf.view(n_items, 8).cpu().double()
    dt = t[:, 1:] - t[:, :-1]
    tot = (t[:, 7] - t[:, 0])
    print(f"S={S} L={L}: items={n_items} wave lifetime median {tot.median():.0f} ticks (s_memtime @100MHz -> x24 cycles)")
    for k, n in enumerate(NAMES):
        print(f"   {n:16s} median {dt[:, k].median():8.0f}  mean {dt[:, k].mean():8.0f}  share {dt[:, k].sum() / tot.sum():.3f}")
