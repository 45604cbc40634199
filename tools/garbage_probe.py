"""Does any kernel of a CACNF / STLT training step read memory nobody wrote?  Run the same seeded steps (a) plain, (b) with every cached block
of torch's allocator and the package's shared scratch buffers filled with NaN between steps, (c) the same with a large finite value."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
import test_train_context_gpu as T


def poison(value):
    torch.cuda.synchronize()
    for buf in list(pkg.ops._SCRATCH.values()) + list(pkg.ops._SK_SCRATCH.values()):
        buf.view(torch.float32)[: buf.numel() // 4].fill_(value)
    junk = []
    try:
        for _ in range(64):
            junk.append(torch.full((16 << 20,), value, device="cuda"))  # 64 MB each
    except RuntimeError:
        pass
    for mb in (1, 2, 4, 8, 16, 32):
        for _ in range(16):
            junk.append(torch.full((mb << 18,), value, device="cuda"))
    del junk
    torch.cuda.synchronize()


def run(make, fusion, n, value):
    torch.manual_seed(0)
    m = make(pkg)
    tr = T._trainer(pkg, m)
    for b in T._batches(pkg, fusion, n):
        if value is not None:
            poison(value)
        tr.step(b)
    torch.cuda.synchronize()
    return [p.detach().clone() for p in m.parameters()]


for name, make, fusion, n in (("stlt", T._stlt, False, 32), ("cacnf", T._cacnf, True, 8)):
    base = run(make, fusion, n, None)
    again = run(make, fusion, n, None)
    print(name, "plain twice identical:", all(torch.equal(a, b) for a, b in zip(base, again)))
    for value in (float("nan"), 1e30, 0.0):
        got = run(make, fusion, n, value)
        nan = sum(int(torch.isnan(g).sum()) for g in got)
        diff = max(float((a - b).abs().max()) for a, b in zip(base, got) if not torch.isnan(b).any()) if nan == 0 else float("nan")
        worst = [k for (k, _), a, b in zip(make(pkg).named_parameters(), base, got) if not torch.equal(a, b)]
        print(f"{name} poisoned with {value}: identical={all(torch.equal(a, b) for a, b in zip(base, got))} NaNs={nan} max|diff|={diff} differing params={len(worst)} e.g. {worst[:4]}")
