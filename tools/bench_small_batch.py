#!/usr/bin/env python3
"""Forward latency / throughput of cfg2 at small per-GPU batches (the reference's default batch size is 64,
parser.py:92-96), plus per-shape GEMM rates with and without stream-K scratch."""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
dev = torch.device("cuda")

def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

out = {"gemm": [], "forward": []}
for (M, N, K) in [(2048, 2304, 768), (2048, 768, 768), (2048, 3072, 768), (2048, 768, 3072), (14336, 2304, 768), (14336, 768, 768),
                  (14336, 3072, 768), (14336, 768, 3072), (4096, 768, 768), (8192, 768, 3072), (16896, 768, 768), (229376, 768, 768)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    t0 = timeit(lambda: pkg.ops.linear(x, w, b, out=y))
    with pkg.ops.gemm_scratch():
        t1 = timeit(lambda: pkg.ops.linear(x, w, b, out=y))
    fl = 2.0 * M * N * K
    out["gemm"].append({"M": M, "N": N, "K": K, "tiles": -(-M // 256) * -(-N // 128), "plain_tflops": round(fl / t0 / 1e12, 1), "streamk_tflops": round(fl / t1 / 1e12, 1)})
c = pkg.synth.CONFIGS["cfg2"]
model = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("cfg2")))
model.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model.train(False).to(dev)
for B in (1, 8, 64, 128, 256):
    batch = {k: v.to(dev) for k, v in pkg.synth.make_batch(B, c["T"], c["N"], seed=3).items()}
    with torch.no_grad():
        t = timeit(lambda: model(batch)["stlt"])
    out["forward"].append({"B": B, "ms": round(t * 1e3, 3), "clips_per_s": round(B / t, 1)})
print(json.dumps(out))
