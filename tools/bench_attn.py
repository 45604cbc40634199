#!/usr/bin/env python3
"""Timing of stlt_attn_core_fwd on the STLT attention shapes (GPU box only): GB/s of algorithmic bytes
(read packed QKV + write ctx + kpm byte) against the 8 TB/s HBM roofline."""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def run(S, L, H, causal, iters):
    d = 64 * H
    dev = "cuda"
    qkv = torch.randn(S, L, 3 * d, device=dev)
    kpm = torch.rand(S, L, device=dev) < 0.2
    kpm[:, 0] = False
    for _ in range(3):
        pkg.ops.attn_core(qkv, kpm, causal, H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        pkg.ops.attn_core(qkv, kpm, causal, H)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    by = S * L * (16.0 * d + 1)
    print(f"S={S:7d} L={L:3d} H={H} causal={int(causal)}  {us:9.1f} us  {by/us/1e3:8.1f} GB/s  ({by/us/1e3/8000:.3f} of 8 TB/s)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batches", type=int, nargs="*", default=[256, 1024])
    args = ap.parse_args()
    for B in args.batches:
        run(B, 32, 12, True, args.iters)        # cfg2 temporal
        run(B, 33, 12, True, args.iters)        # the released checkpoints' layout: 32 + 1 frames (datasets.py:97-113)
        run(B, 17, 12, True, args.iters)        # the parser's default: 16 + 1 frames
        run(B * 32, 7, 12, False, args.iters)   # cfg2 spatial
    for B in (16, 64, 256):
        run(B, 64, 12, True, args.iters)        # cfg4 temporal
        run(B * 64, 36, 12, False, args.iters)  # cfg4 spatial


if __name__ == "__main__":
    main()
