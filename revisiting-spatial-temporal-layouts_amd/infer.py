"""Inference loop for the STLT path: the counterpart of the reference's `inference()` (src/inference.py:15-85).

Step semantics reproduced: eval mode, no_grad, per batch `move_batch_to_device` -> `model(batch)` -> evaluator
update; here every rank runs its own shard of each batch and the top-1/top-5 counters (reference
src/utils/evaluation.py:21-34 keeps two integer counters per logit head) are summed over ranks at the end.  The
counters stay on the device: the reference's per-batch `.cpu()` synchronisation (evaluation.py:25-30) is gone.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional

import torch

from . import dist as D


def topk_counts(logits: torch.Tensor, labels: torch.Tensor, ks=(1, 5)) -> torch.Tensor:
    """Number of rows whose label is among the k largest logits, for each k. -> int64 tensor (len(ks),)
    Device logits with the reference's ks = (1, 5) go through the library's counting kernel (include/stlt_hip.h:
    stlt_eval_topk, no torch kernels); anything else — CPU tensors in the not-gpu tests, other ks — takes the torch form."""
    if logits.is_cuda and tuple(ks) == (1, 5) and logits.dim() == 2 and logits.shape[0] > 0:
        from . import _lib as L
        lib = L.load()
        x = logits if (logits.dtype == torch.float32 and logits.stride(-1) == 1) else logits.float().contiguous()
        y = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        counts = torch.zeros(2, dtype=torch.int64, device=logits.device)
        with torch.cuda.device(logits.device):
            L.check(lib.stlt_eval_topk(x.data_ptr(), x.stride(0), y.data_ptr(), x.shape[0], x.shape[1], counts.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "stlt_eval_topk")
        return counts
    kmax = min(max(ks), logits.shape[1])
    top = logits.topk(kmax, dim=1).indices  # (n, kmax)
    hit = top == labels.view(-1, 1)
    return torch.stack([hit[:, : min(k, kmax)].any(dim=1).sum() for k in ks]).to(torch.int64)


@torch.no_grad()
def run_inference(model, batches: Iterable[Dict[str, torch.Tensor]], device, rank: int = 0, world: int = 1,
                  forward: Optional[Callable] = None, collect_logits: bool = False) -> Dict[str, object]:
    """Evaluate `model` over an iterable of collated batches (each holding `labels`), sharding every batch over ranks.

    `forward` defaults to `model(batch)["stlt"]`; tests on CPU pass a stand-in.  Returns top-1 / top-5 accuracy in
    percent (rounded as the reference logs them, inference.py:83-84), the clip count and optionally all logits.
    """
    if hasattr(model, "train"):
        model.train(False)
    counts = torch.zeros(3, dtype=torch.int64, device=device)  # top1, top5, n
    kept = []
    for batch in batches:
        n_total = batch["categories"].shape[0]
        mine = D.shard_batch(batch, rank, world)
        mine = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in mine.items()}
        if mine["categories"].shape[0] > 0:
            logits = forward(mine) if forward is not None else model(mine)["stlt"]
            c = topk_counts(logits, mine["labels"])
            counts[:2] += c
            counts[2] += logits.shape[0]
        else:
            logits = torch.zeros(0, 1, device=device)
        if collect_logits:
            width = torch.tensor([logits.shape[1] if logits.shape[0] else 0], device=device)
            if world > 1:
                torch.distributed.all_reduce(width, op=torch.distributed.ReduceOp.MAX)
            if logits.shape[0] == 0:
                logits = torch.zeros(0, int(width.item()), device=device)
            kept.append(D.gather_rows(logits.float(), n_total, world).cpu())
    D.all_reduce_sum_(counts, world)
    n = max(int(counts[2].item()), 1)
    out = {"top1_accuracy": round(100.0 * counts[0].item() / n, 2), "top5_accuracy": round(100.0 * counts[1].item() / n, 2),
           "num_clips": int(counts[2].item())}
    if collect_logits:
        out["logits"] = torch.cat(kept, dim=0) if kept else torch.zeros(0, 0)
    return out
