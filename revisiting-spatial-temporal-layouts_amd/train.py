"""Training loop for the STLT path: the counterpart of the reference's `train()` step (src/train.py:102-135).

Step semantics reproduced (defaults of src/utils/parser.py:92-132): AdamW(lr 5e-5) over two parameter groups — no
weight decay for 1-D parameters and names ending in ".bias", weight decay 1e-3 for the rest
(src/utils/train_inference_utils.py:37-54) — a warm-up/linear-decay LambdaLR stepped every batch (:20-34),
`Criterion` = CrossEntropyLoss ("something") or BCEWithLogitsLoss ("action_genome") averaged over the logit heads
(:64-76), gradient clipping at 5.0, and per step: zero_grad -> forward -> loss -> backward -> clip -> step -> sched.

Forward and backward run in the HIP library (modelling/models.py `_StltTrainFn`).  Loss, clipping and AdamW are stock
torch ops on the GPU, exactly as in the reference harness.  Data parallel: every rank runs the step on its shard of the
global batch and the gradients are averaged with ONE all-reduce (RCCL over xGMI; 344 MB for d=768) between backward
and clipping; the loss is a batch mean, so equal shards reproduce the single-process global batch exactly.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.nn.functional as F

from . import dist as D


def add_weight_decay(model: torch.nn.Module, weight_decay: float) -> List[dict]:
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue  # frozen weights
        (no_decay if (param.dim() == 1 or name.endswith(".bias")) else decay).append(param)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def linear_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int):
    def lr_lambda(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))

    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


def criterion(logits: Dict[str, torch.Tensor], labels: torch.Tensor, dataset_name: str = "something") -> torch.Tensor:
    fn = F.cross_entropy if dataset_name == "something" else F.binary_cross_entropy_with_logits
    return sum(fn(v, labels) for v in logits.values()) / len(logits)


def allreduce_gradients(model: torch.nn.Module, world: int) -> None:
    """Average the gradients over ranks with one flat all-reduce (parameters without a gradient — the dead
    `encoder_layer` copy, unused `score_embeddings`, frozen weights — are skipped on every rank alike)."""
    if world == 1:
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    flat = torch._utils._flatten_dense_tensors(grads)
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    flat.div_(world)
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)


class Trainer:
    def __init__(self, model, dataset_name: str = "something", learning_rate: float = 5e-5, weight_decay: float = 1e-3,
                 clip_val: float = 5.0, warmup_steps: int = 0, total_steps: int = 1, rank: int = 0, world: int = 1):
        self.model, self.dataset_name, self.clip_val = model, dataset_name, clip_val
        self.rank, self.world = rank, world
        self.optimizer = torch.optim.AdamW(add_weight_decay(model, weight_decay), lr=learning_rate)
        self.scheduler = linear_schedule_with_warmup(self.optimizer, warmup_steps, total_steps)

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
        """One optimisation step on this rank's shard of the global batch (train.py:119-135)."""
        self.model.train(True)
        self.optimizer.zero_grad()
        logits = self.model(batch)
        loss = criterion(logits, batch["labels"], self.dataset_name)
        loss.backward()
        allreduce_gradients(self.model, self.world)
        grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_val)
        self.optimizer.step()
        self.scheduler.step()
        return {"loss": loss.detach(), "grad_norm": grad_norm.detach()}

    def fit(self, batches: Iterable[Dict[str, torch.Tensor]], device) -> List[Dict[str, float]]:
        """Run over an iterable of GLOBAL batches, each rank taking its contiguous shard."""
        log = []
        for batch in batches:
            mine = D.shard_batch(batch, self.rank, self.world)
            mine = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in mine.items()}
            out = self.step(mine)
            if self.world > 1:  # report the global-batch mean loss
                l = out["loss"].clone()
                torch.distributed.all_reduce(l)
                out["loss"] = l / self.world
            log.append({k: float(v) for k, v in out.items()})
        return log
