#!/usr/bin/env python3
"""Diagnostic for csrc/gemm_bf16x3.hip built with -DSTLT_X3_STAMP=1 (STLT_HIP_LIB=build/variants/libstlt_hip_x3stamp.so):
shader-clock cycles per k-step of every wave and the share spent at the step barrier, MFMA waves vs producer waves."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
lib = pkg._lib.load()
M, N, K = 229376, 768, 3072
if len(sys.argv) > 3:
    M, N, K = map(int, sys.argv[1:4])
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
scratch = torch.empty(int(lib.stlt_gemm_scratch_bytes()), dtype=torch.uint8, device="cuda")  # lent as inside the whole-path calls: the weight is then cut once per launch
pkg._lib.check(lib.stlt_gemm_set_scratch(scratch.data_ptr(), scratch.numel()), "stlt_gemm_set_scratch")
pkg.ops.set_gemm_split_bf16(6)
for _ in range(2):
    pkg.ops.linear(x, w, b, out=y)
buf = torch.zeros(max(int(lib.stlt_debug_buffer_bytes()) // 8, 20480), dtype=torch.int64, device="cuda")
lib.stlt_debug_set_buffer(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); pkg.ops.linear(x, w, b, out=y); e1.record(); torch.cuda.synchronize()
lib.stlt_debug_set_buffer(None)
pkg.ops.set_gemm_split_bf16(0)
ms = e0.elapsed_time(e1)
t = buf[: 256 * 12 * 4].view(256, 12, 4).cpu().double()
steps = t[:, :, 2].clamp(min=1)
tot, bar = t[:, :, 0] / steps, t[:, :, 1] / steps
print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us, {2.0*M*N*K/ms/1e9:.1f} TFLOP/s-equivalent; k-steps per workgroup median {steps[:,0].median():.0f}")
print(f"implied shader clock: {tot[:, 0].median() * steps[:, 0].median() / (ms * 1e3):.0f} cycles/us")
for name, sl in (("MFMA waves 0-7", slice(0, 8)), ("producer waves 8-11", slice(8, 12))):
    print(f"  {name:20s} cycles per k-step: total {tot[:, sl].median():7.0f}   at the barrier {bar[:, sl].median():7.0f}  ({(bar[:, sl] / tot[:, sl]).median():.3f})"
          f"   own work {(tot[:, sl] - bar[:, sl]).median():7.0f}")
for wv in range(12):
    print(f"    wave {wv:2d}: total {tot[:, wv].median():7.0f} barrier {bar[:, wv].median():7.0f}")
ph = buf[256 * 12 * 4: 256 * 12 * 4 + 256 * 4 * 4].view(256, 4, 4).cpu().double() / steps[:, 8:12, None]
if float(ph.sum()) > 0:
    # every phase stamp reads s_memtime, whose wait (lgkmcnt(0)) also waits for the wave's LDS stores: the phases include
    # LDS completion latency (~1000 cycles for one ds_write under the MFMA waves' read traffic), the stamped build runs ~10 % slower
    print("  producer phases, cycles per k-step (median): other/waits %.0f  cut %.0f  plane stores %.0f  load issue %.0f" % tuple(ph[:, :, k].median() for k in range(4)))
ph2 = buf[256 * 12 * 4 + 256 * 16: 256 * 12 * 4 + 256 * 16 + 256 * 8].view(256, 4, 2).cpu().double() / steps[:, 8:12, None]
if float(ph2.sum()) > 0:
    print("  of other: bias store incl. its wait for the carried load %.0f   next-tile origin + bias load issue %.0f" % (ph2[:, :, 0].median(), ph2[:, :, 1].median()))
