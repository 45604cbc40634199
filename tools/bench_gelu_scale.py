"""FFN1-shaped GEMM + GELU at several input scales: the library erff's cost depends on how many lanes of a wave leave |z| < 1."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
M, N, K = 229376, 3072, 768
g = torch.Generator(device="cuda").manual_seed(0)
w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
b = torch.rand(N, device="cuda", generator=g)
y = torch.empty(M, N, device="cuda")
for scale in (1.0, 3.0, 6.0, 12.0):
    x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1) * scale
    for _ in range(2): pkg.ops.linear(x, w, b, act=1, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): pkg.ops.linear(x, w, b, act=1, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    pre = (x[:4096] @ w.t() + b)
    print(f"scale {scale:5.1f}: pre-activation std {pre.std().item():.2f}, share |z|>1 {((pre.abs() / 2 ** 0.5) > 1).float().mean().item():.3f}, {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
