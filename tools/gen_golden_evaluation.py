"""Capture golden metrics from the reference's evaluators (src/utils/evaluation.py) -> tests/golden/evaluation.npz.

Runs in the build container only (imports /root/reference).  numpy 2.x removed `np.NINF`, which the reference's
`charades_map` uses (evaluation.py:131); the alias is restored here before the import, nothing else is touched.
"""
import sys

import numpy as np
import torch

if not hasattr(np, "NINF"):
    np.NINF = -np.inf
sys.path.insert(0, "/root/reference/src")
from utils.evaluation import EvaluatorActionGenome, EvaluatorSomething, charades_map  # noqa: E402

rng = np.random.Generator(np.random.PCG64(77))
out = {}
# Something-Else style: two logit heads, three uneven batches
n, c = 90, 174
la = rng.standard_normal((n, c)).astype(np.float32)
lb = rng.standard_normal((n, c)).astype(np.float32)
labels = rng.integers(0, c, size=n)
for i in range(0, n, 3):  # plant correct answers so the counters are not near zero
    la[i, labels[i]] += 4.0
    lb[i, labels[i]] += 2.0
ev = EvaluatorSomething(n, c, ("stlt", "caf"))
for lo, hi in ((0, 32), (32, 64), (64, 90)):
    ev.process({"stlt": torch.from_numpy(la[lo:hi]), "caf": torch.from_numpy(lb[lo:hi])}, torch.from_numpy(labels[lo:hi]))
m = ev.evaluate()
out.update(sth_logits_a=la, sth_logits_b=lb, sth_labels=labels,
           sth_metrics=np.array([m["stlt_top1_accuracy"], m["stlt_top5_accuracy"], m["caf_top1_accuracy"], m["caf_top5_accuracy"]]))
# Action Genome style: multi-label, some clips without any action, one class without positives
n, c = 120, 157
lg = (rng.standard_normal((n, c)) * 2).astype(np.float32)
gt = (rng.random((n, c)) < 0.06).astype(np.float32)
gt[5] = 0
gt[17] = 0
gt[:, 11] = 0
ev = EvaluatorActionGenome(n, c, ("stlt",))
for lo, hi in ((0, 50), (50, 120)):
    ev.process({"stlt": torch.from_numpy(lg[lo:hi])}, torch.from_numpy(gt[lo:hi]))
m_ap, w_ap, aps = charades_map(ev.predictions, ev.ground_truths)
out.update(ag_logits=lg, ag_truths=gt, ag_map=np.array(m_ap), ag_wap=w_ap, ag_aps=aps)
# the same with the empty class removed, so the mean is finite
keep = [j for j in range(c) if j != 11]
m2, w2, a2 = charades_map(ev.predictions[:, keep], ev.ground_truths[:, keep])
out.update(ag_map_finite=np.array(m2), ag_aps_finite=a2)
np.savez_compressed("tests/golden/evaluation.npz", **out)
print("sth", m, "ag map", m_ap, "finite", m2)
