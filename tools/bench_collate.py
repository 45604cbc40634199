#!/usr/bin/env python3
"""f-1 / f-4 measurement: collating a batch of per-video layout dicts and feeding an evaluator, the reference's way
(pad on the CPU, copy the padded tensors to the GPU; `.cpu()` on the logits of every batch) against this repo's way
(one ragged copy per field + one kernel; counters / score tables stay on the device).  The CPU side uses the oracle's
restatement of the reference collater and evaluator — this is a measurement tool, not the product path."""
import importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
from oracle import collate_oracle as CO  # noqa: E402
E = importlib.import_module("revisiting-spatial-temporal-layouts_amd.utils.evaluation")
dev = torch.device("cuda")


def timeit(fn, n=10, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def video_samples(dataset, n, N, T, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for i in range(n):
        nf = int(rng.integers(T // 2, T + 1))
        b = pkg.synth.make_batch(1, nf, N, dataset=dataset, seed=seed * 1000 + i, with_scores=True, min_len=nf)
        out.append({"video_id": f"v{i}", "categories": b["categories"][0], "boxes": b["boxes"][0], "scores": b["scores"][0],
                    "frame_types": b["frame_types"][0], "lengths": torch.tensor(nf), "labels": torch.tensor(int(rng.integers(0, 174)))})
    return out


res = []
for dataset, T, N in (("something", 32, 7), ("action_genome", 64, 36)):
    for B in (64, 1024):
        samples = video_samples(dataset, B, N, T, 3)
        dc = pkg.collate.DeviceCollater(dataset, dev)

        def ref_way():
            b = CO.collate(samples, dataset)
            return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}

        t_ref, t_dev = timeit(ref_way, n=5, w=2), timeit(lambda: dc(samples), n=5, w=2)
        res.append({"what": "collate", "dataset": dataset, "B": B, "T": T, "N": N, "reference_way_ms": round(t_ref * 1e3, 2),
                    "device_collater_ms": round(t_dev * 1e3, 2)})
logits = torch.randn(1024, 174, device=dev)
labels = torch.randint(0, 174, (1024,), device=dev)


def ref_eval():
    top = logits.cpu()
    return (top.argmax(-1) == labels.cpu()).sum().item(), (top.topk(5).indices == labels.cpu().unsqueeze(1)).any(1).sum().item()


ev = E.EvaluatorSomething(10 ** 9, 174, ("stlt",))
res.append({"what": "evaluator.process, 1024 clips", "reference_way_ms": round(timeit(ref_eval) * 1e3, 3),
            "device_ms": round(timeit(lambda: ev.process({"stlt": logits}, labels)) * 1e3, 3)})
ag_logits, ag_truth = torch.randn(1814, 157, device=dev), (torch.rand(1814, 157, device=dev) < 0.06).float()
t_map = timeit(lambda: E.charades_map(ag_logits.sigmoid(), ag_truth)[0].item(), n=5, w=1)
res.append({"what": "Charades mAP, 1814 clips x 157 classes (sort + cumulative sums on the device)", "device_ms": round(t_map * 1e3, 3)})
print(json.dumps(res, indent=1))
