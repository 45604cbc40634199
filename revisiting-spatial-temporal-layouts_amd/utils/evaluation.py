"""Evaluators that keep their state on the device.

Same class surface as the reference's `src/utils/evaluation.py:5-138` (`EvaluatorSomething`, `EvaluatorActionGenome`,
`evaluators_factory`; `reset / process / evaluate / is_best`), but `process()` never copies logits to the host: the
reference's `.cpu()` per batch (evaluation.py:25-30, 79-82) is an implicit device synchronisation in the inference
loop.  Counters / score tables live on the logits' device and are read back once, in `evaluate()`.

Charades mAP (evaluation.py:100-132): per class, sort the clips by descending score, precision at every positive,
mean over positives; clips with no ground-truth action are pushed to the end (score = -inf) first; classes without a
positive give NaN, and the mean over classes then is NaN exactly as numpy's `np.mean` gives the reference.  Here it is
one batched descending sort over the (clips, classes) table plus two cumulative sums, in float64 like the reference.
Ties: the reference's `np.argsort` is unstable, so its result for tied scores is unspecified; this one is stable.

Device tensors go through hand-written HIP kernels (csrc/evalk.hip, include/stlt_hip.h: `stlt_eval_topk` — one wave
per clip counts the classes that beat the label, hits accumulate in int64 device counters; `stlt_eval_average_precision`
— one workgroup per class sorts its column in LDS and sums the precision at every positive in a fixed order).  CPU
tensors (the not-gpu tests, and a reference-style host loop) take the same arithmetic as torch ops.

Multi-rank: `process()` is fed the rank's shard; `evaluate()` sums the counters / gathers the score tables over the
default process group when one is initialised.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class EvaluatorSomething:
    """Top-1 / top-5 accuracy per logit head (reference evaluation.py:5-58)."""

    def __init__(self, total_instances: int, total_classes: int, logit_names: Tuple[str, ...]):
        self.total_instances = total_instances
        self.total_classes = total_classes
        self.logit_names = tuple(logit_names)
        self.best_acc = 0.0
        self.reset()

    def reset(self):
        self._counts = None  # (len(logit_names), 2) int64 on the logits' device, made on first process()

    def process(self, logits: Dict[str, torch.Tensor], labels: torch.Tensor):
        assert len(logits) == len(self.logit_names)
        dev = logits[self.logit_names[0]].device
        if self._counts is None:
            self._counts = torch.zeros(len(self.logit_names), 2, dtype=torch.int64, device=dev)
        labels = labels.to(dev)
        if dev.type == "cuda":  # the HIP path: no torch kernels, no synchronisation
            from .. import _lib as L
            lib = L.load()
            y = labels.to(torch.int64).contiguous()
            with torch.cuda.device(dev):
                for i, name in enumerate(self.logit_names):
                    x = logits[name]
                    if x.dtype != torch.float32 or x.stride(-1) != 1 or x.dim() != 2:
                        x = x.float().contiguous()
                    L.check(lib.stlt_eval_topk(x.data_ptr(), x.stride(0), y.data_ptr(), x.shape[0], x.shape[1],
                                               self._counts[i].data_ptr(), torch.cuda.current_stream().cuda_stream), "stlt_eval_topk")
            return
        labels = labels.view(-1, 1)
        for i, name in enumerate(self.logit_names):
            top = logits[name].topk(k=min(5, logits[name].shape[1]), dim=1).indices
            hit = top == labels
            self._counts[i, 0] += hit[:, 0].sum()
            self._counts[i, 1] += hit.any(dim=1).sum()

    @property
    def corrects(self) -> Dict[str, int]:
        """The reference's public counter dict (evaluation.py:16-20), read back from the device."""
        c = self._total_counts()
        out = {}
        for i, name in enumerate(self.logit_names):
            out[f"{name}_top1"] = int(c[i, 0])
            out[f"{name}_top5"] = int(c[i, 1])
        return out

    def _total_counts(self):
        if self._counts is None:
            return torch.zeros(len(self.logit_names), 2, dtype=torch.int64)
        c = self._counts.clone()
        if _world() > 1:
            dist.all_reduce(c)
        return c.cpu()

    def evaluate(self) -> Dict[str, float]:
        c = self._total_counts()
        metrics = {}
        for i, name in enumerate(self.logit_names):
            metrics[f"{name}_top1_accuracy"] = int(c[i, 0]) / self.total_instances
            metrics[f"{name}_top5_accuracy"] = int(c[i, 1]) / self.total_instances
        return metrics

    def is_best(self) -> bool:
        metrics = self.evaluate()
        cur = sum(metrics.values()) / len(metrics)
        if cur > self.best_acc:
            self.best_acc = cur
            return True
        return False


def average_precisions(scores: torch.Tensor, truths: torch.Tensor) -> torch.Tensor:
    """Per-class average precision (reference `map`, evaluation.py:100-124).  scores, truths: (clips, classes) float64.
    -> (classes,) float64, NaN where a class has no positive."""
    order = torch.sort(scores, dim=0, descending=True, stable=True).indices
    tp = torch.gather(truths, 0, order) == 1
    tpc = torch.cumsum(tp.to(torch.float64), dim=0)
    rank = torch.arange(1, scores.shape[0] + 1, dtype=torch.float64, device=scores.device).view(-1, 1)
    prec = tpc / rank  # t_pcs / (f_pcs + t_pcs): the denominator is the 1-based rank
    n_pos = tp.sum(dim=0)
    ap = (prec * tp).sum(dim=0) / n_pos.to(torch.float64)
    return torch.where(n_pos > 0, ap, torch.full_like(ap, float("nan")))


def _hip_ap_can_take(scores: torch.Tensor) -> bool:
    from .. import _lib as L
    n = scores.shape[0]
    if n == 0 or n > int(L.load().stlt_eval_max_clips()):
        return False
    return scores.dtype == torch.float32 or bool((scores.to(torch.float32).to(scores.dtype) == scores).all())


def _average_precisions_hip(scores: torch.Tensor, truths: torch.Tensor) -> torch.Tensor:
    """`stlt_eval_average_precision` on device tables (the empty-clip rule is applied inside)."""
    from .. import _lib as L
    lib = L.load()
    n, C = scores.shape
    if n > int(lib.stlt_eval_max_clips()):
        raise L.StltHipError(f"charades_map: {n} clips exceed the {int(lib.stlt_eval_max_clips())} the device kernel sorts per class")
    s32 = scores.to(torch.float32).contiguous()  # the scores are fp32 sigmoids widened to float64 (evaluation.py:79-81): exact
    t32 = truths.to(torch.float32).contiguous()
    ap = torch.empty(C, dtype=torch.float64, device=scores.device)
    pos = torch.empty(C, dtype=torch.float64, device=scores.device)
    scratch = torch.empty(n, dtype=torch.uint8, device=scores.device)
    with torch.cuda.device(scores.device):
        L.check(lib.stlt_eval_average_precision(s32.data_ptr(), t32.data_ptr(), n, C, ap.data_ptr(), pos.data_ptr(), scratch.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), "stlt_eval_average_precision")
    return ap, pos


def charades_map(scores: torch.Tensor, truths: torch.Tensor):
    """(mAP, weighted AP per class, AP per class) as the reference's `charades_map` (evaluation.py:127-132)."""
    if scores.is_cuda and _hip_ap_can_take(scores):
        aps, pos = _average_precisions_hip(scores, truths)
        w_ap = aps * pos / pos.sum()  # gt.sum(axis=0) = positives per class (multi-hot truths)
        return aps.mean(), w_ap, aps
    # Tables the sorting kernel does not take — more clips than one workgroup sorts in LDS, an empty table, or float64 scores
    # that fp32 cannot hold exactly (narrowing them could reorder near-ties) — keep the batched torch form of the same
    # arithmetic on whatever device the table lives on.
    scores = scores.to(torch.float64).clone()
    truths = truths.to(torch.float64)
    empty = truths.sum(dim=1) == 0
    scores[empty] = float("-inf")
    aps = average_precisions(scores, truths)
    w_ap = aps * truths.sum(dim=0) / truths.sum()
    return aps.mean(), w_ap, aps


class EvaluatorActionGenome:
    """Charades-style mAP over sigmoid scores of the `stlt` head (reference evaluation.py:61-97)."""

    def __init__(self, total_instances: int, total_classes: int, logit_names: Tuple[str, ...]):
        self.total_instances = total_instances
        self.total_classes = total_classes
        self.logit_names = tuple(logit_names)
        self.best_mean_average_precision = 0.0
        self.reset()

    def reset(self):
        self.index = 0
        self.predictions = None  # (total_instances, classes) float64 on the logits' device
        self.ground_truths = None

    def process(self, logits: Dict[str, torch.Tensor], labels: torch.Tensor):
        x = logits["stlt"]
        if self.predictions is None:
            self.predictions = torch.zeros(self.total_instances, self.total_classes, dtype=torch.float64, device=x.device)
            self.ground_truths = torch.zeros_like(self.predictions)
        size = x.shape[0]
        if self.index + size > self.total_instances:
            raise IndexError(f"EvaluatorActionGenome: {self.index + size} clips processed, tables hold {self.total_instances}")
        if x.is_cuda:  # one small kernel: fp32 sigmoid + both table writes (include/stlt_hip.h: stlt_eval_store_sigmoid)
            from .. import _lib as L
            lib = L.load()
            xs = x if (x.dtype == torch.float32 and x.dim() == 2 and x.stride(-1) == 1) else x.float().contiguous()
            ys = labels.to(device=x.device, dtype=torch.float32).contiguous()
            with torch.cuda.device(x.device):
                L.check(lib.stlt_eval_store_sigmoid(xs.data_ptr(), xs.stride(0), ys.data_ptr(), size, self.total_classes, self.predictions.data_ptr(),
                                                    self.ground_truths.data_ptr(), self.index, torch.cuda.current_stream().cuda_stream),
                        "stlt_eval_store_sigmoid")
        else:
            self.predictions[self.index : self.index + size] = x.float().sigmoid()  # fp32 sigmoid, widened: evaluation.py:79-81
            self.ground_truths[self.index : self.index + size] = labels.to(x.device)
        self.index += size

    def _tables(self):
        p, g = self.predictions[: self.index], self.ground_truths[: self.index]
        w = _world()
        if w > 1:  # shards may differ in size by one row: pad to the maximum, gather, trim
            n = torch.tensor([self.index], device=p.device)
            sizes = [torch.zeros_like(n) for _ in range(w)]
            dist.all_gather(sizes, n)
            m = int(max(s.item() for s in sizes))
            pad = lambda t: torch.cat([t, t.new_zeros(m - t.shape[0], t.shape[1])])  # noqa: E731
            ps = [p.new_zeros(m, p.shape[1]) for _ in range(w)]
            gs = [p.new_zeros(m, p.shape[1]) for _ in range(w)]
            dist.all_gather(ps, pad(p))
            dist.all_gather(gs, pad(g))
            p = torch.cat([t[: int(s.item())] for t, s in zip(ps, sizes)])
            g = torch.cat([t[: int(s.item())] for t, s in zip(gs, sizes)])
        if p.shape[0] < self.total_instances:  # rows never filled stay zero, as in the reference's preallocated arrays
            fill = p.new_zeros(self.total_instances - p.shape[0], p.shape[1])
            p, g = torch.cat([p, fill]), torch.cat([g, fill])
        return p, g

    def evaluate(self) -> Dict[str, float]:
        if self.predictions is None:
            return {"map": float("nan")}
        p, g = self._tables()
        m_ap, _, _ = charades_map(p, g)
        return {"map": float(m_ap.item())}

    def is_best(self) -> bool:
        metrics = self.evaluate()
        if metrics["map"] > self.best_mean_average_precision:
            self.best_mean_average_precision = metrics["map"]
            return True
        return False


evaluators_factory = {"something": EvaluatorSomething, "action_genome": EvaluatorActionGenome}
