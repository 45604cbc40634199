"""Run one nn.Linear shape 50 times, with or without the stream-K scratch lent, for profiling under rocprofv3:

    tools/prof_kernels.sh sk tools/sk_probe.py <M> <N> <K> <0|1>
"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
dev = torch.device("cuda")
M, N, K = [int(v) for v in sys.argv[1:4]]
sk = sys.argv[4] == "1"
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
def run():
    for _ in range(50): pkg.ops.linear(x, w, b, out=y)
    torch.cuda.synchronize()
if sk:
    with pkg.ops.gemm_scratch(): run()
else:
    run()
