#!/usr/bin/env python3
"""Diagnostic: per-workgroup start/end times of the persistent GEMM (100 MHz s_memrealtime), grouped by XCD."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
lib = pkg._lib.load()
M, N, K = 229376, 768, 3072
RES = "--residual" in sys.argv  # the residual-add instantiation (out-proj / FFN2 of a post-norm layer): y = x·Wᵀ + r through stlt_gemm's add-source
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if len(args) >= 3: M, N, K = map(int, args[:3])
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
r = torch.randn(M, N, device="cuda") if RES else None
stream = torch.cuda.current_stream().cuda_stream
def product():
    if RES:
        pkg._lib.check(lib.stlt_gemm(0, 0, x.data_ptr(), K, w.data_ptr(), K, r.data_ptr(), N, y.data_ptr(), N, 0, M, N, K, 1, stream), "stlt_gemm")
    else:
        pkg.ops.linear(x, w, b, out=y)
for _ in range(2): product()
buf = torch.zeros(4 * 4096 + 8192, dtype=torch.int64, device="cuda")
lib.stlt_debug_set_buffer(buf.data_ptr())
product(); torch.cuda.synchronize()
lib.stlt_debug_set_buffer(None)
print(f"shape M={M} N={N} K={K}  epilogue: {'+ residual (add-source)' if RES else '+ bias'}")
G = min(((M + 255) // 256) * ((N + 127) // 128), 256)
WAVES = 8
t = buf[: 4 * G].view(G, 4).cpu()
t0 = t[:, 0].min()
st = (t[:, 0] - t0).double() / 100.0  # us
en = (t[:, 1] - t0).double() / 100.0
ck = buf[4 * G + G * WAVES * 6 + 1024: 4 * G + G * WAVES * 6 + 1024 + 2 * G].view(G, 2).cpu().double()
clk = ((ck[:, 1] - ck[:, 0]) / ((t[:, 1] - t[:, 0]).double() / 100.0)).median() / 1e3  # cycles per us -> GHz
fl = 2.0 * M * N * K
print(f"in-kernel clock {clk:.3f} GHz ; kernel {fl / (en.max() * 1e-6) / 1e12:.1f} TFLOP/s by its own span = {fl / (en.max() * 1e-6) / 1e12 / (clk / 2.4 * 157.3):.3f} of the MFMA rate at that clock")
print(f"workgroups={len(t)} tiles/wg min {t[:,3].min()} max {t[:,3].max()}  kernel span {en.max():.1f} us")
print(f"start: max {st.max():.1f} us ; end: min {en.min():.1f} median {en.median():.1f} max {en.max():.1f} us ; mean idle tail {(en.max()-en).mean():.1f} us")
wg_clk = (ck[:, 1] - ck[:, 0]) / ((t[:, 1] - t[:, 0]).double() / 100.0) / 1e3  # GHz per workgroup
for xcc in sorted(set(t[:, 2].tolist())):
    m = t[:, 2] == xcc
    print(f"  xcc {xcc}: wgs {int(m.sum()):4d} end min {en[m].min():8.1f} med {en[m].median():8.1f} max {en[m].max():8.1f}   shader clock {wg_clk[m].median():.3f} GHz"
          f"   blockIdx&7 = {sorted(set((torch.nonzero(m).flatten() & 7).tolist()))}")
import numpy as np
e = en.numpy()
hist, edges = np.histogram(e, bins=12)
print("end-time histogram:", list(zip([f"{x:.0f}" for x in edges[:-1]], hist.tolist())))
slow = np.nonzero(e > (np.median(e) * 1.1))[0]
print("slow workgroups (blockIdx):", slow[:64].tolist(), "count", len(slow))
if os.environ.get("STLT_GEMM_STAMP"):
    ph = buf[4 * G: 4 * G + G * WAVES * 6].view(G * WAVES, 6).cpu().double()
    names = ["chunks0-2 (48 MFMA)", "wait vmcnt/lgkm", "barrier", "DMA issue", "chunk3 (16 MFMA)", "epilogue"]
    tot = ph.sum(1)
    steps = float(t[:, 3].double().mean()) * (K // 32)
    print(f"per-wave cycles total median {tot.median():.0f}; per k-step:")
    for k, n in enumerate(names):
        print(f"   {n:22s} share {ph[:, k].sum() / tot.sum():.3f}   per-step median {(ph[:, k] / steps).median():8.1f} cycles")
    print("per wave index (median over workgroups, cycles per k-step):")
    pw = ph.view(G, WAVES, 6) / steps
    for wv in range(WAVES):
        print(f"   wave {wv}: " + "  ".join(f"{names[k].split()[0]} {pw[:, wv, k].median():7.1f}" for k in range(6)) + f"  total {pw[:, wv].sum(1).median():8.1f}")
    TR_STEPS, TR_PTS = 6, 12
    tr = buf[4 * G + G * WAVES * 6: 4 * G + G * WAVES * 6 + WAVES * TR_STEPS * TR_PTS].view(WAVES, TR_STEPS, TR_PTS).cpu()
    if int(tr.max()) > 0:
        pts = ["top", "c0 done", "dma0", "c1 done", "dma1", "c2 done", "dma2", "waited", "barrier", "c3 done"]
        t0 = int(tr[:, 0, 0].min())
        print("timeline of workgroup 0 (cycles since the first wave entered k-step 30); one row per wave and k-step:")
        for st in range(TR_STEPS):
            for wv in range(WAVES):
                print(f"  step {30 + st} wave {wv}: " + " ".join(f"{pts[k]}={int(tr[wv, st, k]) - t0:6d}" for k in range(len(pts))))
