"""Training loop for the STLT path: the counterpart of the reference's `train()` step (src/train.py:102-135).

Step semantics reproduced (defaults of src/utils/parser.py:92-132): AdamW(lr 5e-5) over two parameter groups — no
weight decay for 1-D parameters and names ending in ".bias", weight decay 1e-3 for the rest
(src/utils/train_inference_utils.py:37-54) — a warm-up/linear-decay LambdaLR stepped every batch (:20-34),
`Criterion` = CrossEntropyLoss ("something") or BCEWithLogitsLoss ("action_genome") averaged over the logit heads
(:64-76), gradient clipping at 5.0, and per step: zero_grad -> forward -> loss -> backward -> clip -> step -> sched.

Forward and backward run in the HIP library (modelling/models.py `_StltTrainFn`), and so do — on a GPU, the default — the
criterion with its gradient (`fused_criterion`: stlt_loss_fwd_bwd), the gradient norm + clipping factor and AdamW
(`FusedAdamW`: stlt_grad_norm, stlt_adamw_step, straight from the flat gradient buffer the reverse sweep fills);
`Trainer(fused_optimizer=False)` keeps the stock torch ops of the reference harness.  Data parallel: every rank runs the
step on its shard of the global batch and the flat gradient buffer is averaged in place over RCCL (344 MB for d=768), in
two slices of which the first travels while the spatial half of the reverse sweep runs; the loss is a batch mean, so equal
shards reproduce the single-process global batch exactly.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.nn.functional as F

import os

from . import _lib as L
from . import dist as D
from . import ops

_BLOCK_DW_DEFER = os.environ.get("STLT_BLOCK_DW_DEFER", "1") != "0"  # A/B knob: the fusion models' weight gradients per block instead of deferred


def add_weight_decay(model: torch.nn.Module, weight_decay: float) -> List[dict]:
    """The optimizer's two parameter groups (contract of train_inference_utils.py:37-54): group 0, no decay, holds the
    trainable vectors — biases, LayerNorm scales, anything one-dimensional; group 1 holds the trainable matrices."""
    groups = ([], [])
    for name, p in model.named_parameters():
        if p.requires_grad:
            is_matrix = p.dim() > 1 and not name.endswith(".bias")
            groups[int(is_matrix)].append(p)
    return [dict(params=groups[0], weight_decay=0.0), dict(params=groups[1], weight_decay=weight_decay)]


class _WarmupThenLinearDecay:
    """Learning-rate factor of the reference's schedule (train_inference_utils.py:20-34): 0 -> 1 over the warm-up steps,
    then a straight line down to 0 at the last training step, never negative."""

    def __init__(self, warmup_steps: int, total_steps: int):
        self.warmup = max(1, int(warmup_steps))
        self.ramp_steps = int(warmup_steps)
        self.total = int(total_steps)
        self.decay_span = max(1, self.total - self.ramp_steps)

    def __call__(self, step: int) -> float:
        if step < self.ramp_steps:
            return step / self.warmup
        return max(0.0, (self.total - step) / self.decay_span)


def linear_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int):
    return torch.optim.lr_scheduler.LambdaLR(optimizer, _WarmupThenLinearDecay(num_warmup_steps, num_training_steps))


def criterion(logits: Dict[str, torch.Tensor], labels: torch.Tensor, dataset_name: str = "something") -> torch.Tensor:
    fn = F.cross_entropy if dataset_name == "something" else F.binary_cross_entropy_with_logits
    return sum(fn(v, labels) for v in logits.values()) / len(logits)


def fused_criterion(logits: torch.Tensor, labels: torch.Tensor, dataset_name: str = "something", weight: float = 1.0):
    """Loss and d(loss)/d(logits) of one logit head in one native pass (include/stlt_hip.h: stlt_loss_fwd_bwd).
    -> (loss scalar tensor, dlogits): feed `logits.backward(dlogits)`."""
    from . import _lib as L
    lib = L.load()
    B, K = logits.shape
    x = logits.detach().contiguous().float()
    if dataset_name == "something":
        kind, y = 0, labels.to(torch.int64).contiguous()
    else:
        kind, y = 1, labels.to(torch.float32).contiguous()
    out = torch.empty(B + 1, device=x.device, dtype=torch.float32)
    dlogits = torch.empty_like(x)
    L.check(lib.stlt_loss_fwd_bwd(x.data_ptr(), y.data_ptr(), kind, B, K, float(weight), out.data_ptr(), out[B:].data_ptr(),
                                  dlogits.data_ptr(), torch.cuda.current_stream().cuda_stream), "stlt_loss_fwd_bwd")
    return out[B], dlogits


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


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (and its state layout: `step`, `exp_avg`, `exp_avg_sq` per parameter, so state dicts
    are interchangeable) fed from the flat gradient buffer the native reverse sweep fills: gradient norm + clipping
    factor in one reduction, then one update kernel over a chunk table (include/stlt_hip.h: stlt_grad_norm,
    stlt_adamw_step).  Parameters absent from the flat buffer (no gradient this step) are left alone, like torch's
    `if p.grad is None: continue`."""

    CHUNK = 16384

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._layout_key = None
        self._group_step = {}

    def _bind(self, layout, device):
        from . import _lib as L
        import ctypes as C
        import numpy as np
        # a re-bind in the middle of training (another set of parameters received gradients, parameters moved, a second
        # Trainer) must not lose the step count: the host-side counters go into the per-parameter `step` tensors first, from
        # which step_flat() re-seeds them below (after load_state_dict() the counters are empty and the loaded tensors stand)
        self._sync_step_tensors()
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        total = max((off + (n + 3) // 4 * 4 for _, off, n in layout), default=0)
        self._m = torch.zeros(total, device=device, dtype=torch.float32)
        self._v = torch.zeros(total, device=device, dtype=torch.float32)
        self._tables = []  # one chunk table per parameter group (lr / betas / eps are per group)
        for g in self.param_groups:
            rows = []
            for p, off, n in layout:
                if group_of.get(id(p)) is not g:
                    continue
                old = self.state.get(p, {})
                st = self.state[p] = {"step": old.get("step", torch.tensor(0.0)),
                                      "exp_avg": self._m[off: off + n].view_as(p), "exp_avg_sq": self._v[off: off + n].view_as(p)}
                if "exp_avg" in old and old["exp_avg"].data_ptr() != st["exp_avg"].data_ptr():  # loaded from a checkpoint / a previous binding: move it into the flat buffers
                    st["exp_avg"].copy_(old["exp_avg"].to(device)); st["exp_avg_sq"].copy_(old["exp_avg_sq"].to(device))
                for c0 in range(0, n, self.CHUNK):
                    rows.append((p.data_ptr() + 4 * c0, off + c0, min(self.CHUNK, n - c0), float(g["weight_decay"])))
            arr = np.zeros(len(rows), dtype=np.dtype([("param", "<u8"), ("off", "<i8"), ("n", "<i4"), ("wd", "<f4")]))
            for i, r in enumerate(rows):
                arr[i] = r
            assert arr.dtype.itemsize == C.sizeof(L.OptChunk)
            self._tables.append((g, torch.from_numpy(arr.view(np.uint8).copy()).to(device), len(rows)))
        self._scratch = torch.zeros(1024 + 2, device=device, dtype=torch.float32)
        self._group_step = {}  # re-read from the (possibly just loaded) per-parameter step tensors at the next step
        self._layout_key = tuple((id(p), p.data_ptr(), off, n) for p, off, n in layout)

    @torch.no_grad()
    def step_flat(self, flat: torch.Tensor, layout, max_norm: float = 0.0) -> torch.Tensor:
        """One optimisation step from the flat gradient buffer; returns the (pre-clipping) gradient norm, on the device."""
        from . import _lib as L
        lib = L.load()
        key = tuple((id(p), p.data_ptr(), off, n) for p, off, n in layout)
        if key != self._layout_key:
            self._bind(layout, flat.device)
        stream = torch.cuda.current_stream().cuda_stream
        self._opt_called = True  # what torch's wrapped step() records for the LR scheduler's call-order check
        out = self._scratch[1024:]
        L.check(lib.stlt_grad_norm(flat.data_ptr(), flat.numel(), float(max_norm), self._scratch.data_ptr(), out.data_ptr(), stream),
                "stlt_grad_norm")
        for g, table, n_chunks in self._tables:
            if n_chunks == 0:
                continue
            # torch keeps one `step` tensor per parameter; the parameters of a group step together, so the host keeps one
            # integer per group and writes the per-parameter tensors only when a state dict is asked for (state_dict())
            gi = id(g)
            if gi not in self._group_step:
                live = [float(self.state[p]["step"]) for p in g["params"] if "exp_avg" in self.state.get(p, {})]
                if not live:
                    continue
                self._group_step[gi] = int(max(live))
            self._group_step[gi] += 1
            step = self._group_step[gi]
            L.check(lib.stlt_adamw_step(table.data_ptr(), n_chunks, flat.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                        out.data_ptr() if max_norm > 0 else None, float(g["lr"]), float(g["betas"][0]),
                                        float(g["betas"][1]), float(g["eps"]), step, stream), "stlt_adamw_step")
        return out[0]

    def step(self, closure=None):
        raise RuntimeError("FusedAdamW consumes the flat gradient buffer of the native backward: call step_flat(flat, layout)")

    def _sync_step_tensors(self):
        for g in self.param_groups:
            n = getattr(self, "_group_step", {}).get(id(g))
            if n is None:
                continue
            for p in g["params"]:
                st = self.state.get(p)
                if st is not None and "exp_avg" in st:
                    st["step"] = torch.tensor(float(n))  # one tensor per parameter, as torch.optim.AdamW keeps them

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """torch's loader replaces `self.state[p]` with fresh tensors, which no longer alias the flat moment buffers the
        kernel updates: drop the binding so that the next step_flat() re-binds and copies the loaded moments in."""
        super().load_state_dict(state_dict)
        self._layout_key = None
        self._group_step = {}


class BoundFlatGrads:
    """Every trainable parameter's .grad as a view of ONE flat buffer (16-byte aligned slots), for models whose backward is
    composed of autograd Functions (the fusion models): autograd accumulates into the views in place, a step clears the buffer
    with one memset, the data-parallel all-reduce and the fused clip + AdamW (FusedAdamW.step_flat) read it directly.
    Parameters that received no gradient in a step are left out of that step's layout — torch.optim.AdamW skips
    `p.grad is None` the same way (the dead `encoder_layer` copy, unused `score_embeddings`, frozen weights)."""

    def __init__(self, model: torch.nn.Module, context=None):
        self.context = context  # the trainer's ops.TrainContext: what the native backwards of these parameters name while `accumulating`
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.layout_all, off = [], 0
        for p in self.params:
            self.layout_all.append((p, off, p.numel()))
            off += (p.numel() + 3) // 4 * 4
        dev = self.params[0].device
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self._touched = set()
        self._slot = {}
        self.accumulating = False  # True only around the Trainer's own backward: the native backwards then add into the views in place
        for p, o, n in self.layout_all:
            p.grad = self.flat[o: o + n].view_as(p)
            p.register_post_accumulate_grad_hook(lambda q, s=self._touched: s.add(id(q)))
            p._stlt_bound = self  # the native block backwards accumulate straight into the view (ops.grad_targets)
            self._slot[id(p)] = o
        self._layout_cache = (None, None)
        # id(parameter) -> its view of the flat buffer: what the native backwards write into while `accumulating` (a dictionary
        # look-up instead of p.grad + two data_ptr() calls per parameter: 170 parameters of a backbone were 0.8 ms of host time
        # in front of the layout sweep's first launch, and the GPU waited for half of it)
        self._view = {id(p): p.grad for p, _, _ in self.layout_all}

    def view_of(self, p):
        """The parameter's gradient view inside a Trainer step (zero() has just re-bound every .grad), else None."""
        return self._view.get(id(p)) if self.accumulating else None

    def owns(self, p) -> bool:
        o = self._slot.get(id(p))
        return o is not None and p.grad is not None and p.grad.data_ptr() == self.flat.data_ptr() + 4 * o

    def touch(self, p):
        self._touched.add(id(p))

    def zero(self):
        for p, o, n in self.layout_all:  # a caller (or zero_grad(set_to_none=True)) may have dropped a view: bind it again
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                p.grad = self._view[id(p)] = self.flat[o: o + n].view_as(p)
        self.flat.zero_()
        self._touched.clear()

    def layout(self):
        key = frozenset(self._touched)
        if self._layout_cache[0] != key:
            self._layout_cache = (key, [(p, o, n) for p, o, n in self.layout_all if id(p) in key])
        return self._layout_cache[1]


class TransposedWeights:
    """Transposed copies of the model's Linear weights for the input-gradient products of a training step (include/stlt_hip.h:
    stlt_ctx_wt_refresh; csrc/wt_cache.hip): dX = dY·W reads W as it lies 13 - 17 % slower than a forward product reads Wt.  One flat buffer
    holds every copy; `refresh()` rewrites them from the weights as they are NOW and makes them current IN THE TRAINER'S CONTEXT, `clear()`
    withdraws them.  Only backward calls that name that context can read a copy, and the library orders each such call's stream behind the
    transposes (an event recorded at the end of the refresh), so the copies are written on a stream of their own beside the step's forward.
    The Trainer refreshes at the start of every step (after whatever changed the weights: its own optimiser, a checkpoint load, an EMA swap)
    and clears at the end, so nothing outside a step can read a stale copy.  STLT_TRAIN_WT=0 switches it off."""

    MIN_ELEMENTS = 65536  # the 768 x 768 projections and up; embedding tables and the 174-class heads stay as they are

    def __init__(self, model: torch.nn.Module, context):
        self.context = context
        first = next((p for p in model.parameters() if p.is_cuda), None)
        dev = first.device if first is not None else None
        self.params = [p for p in model.parameters()
                       if p.requires_grad and p.is_cuda and p.device == dev and p.dim() == 2 and p.dtype == torch.float32
                       and p.numel() >= self.MIN_ELEMENTS and p.shape[0] % 32 == 0 and p.shape[1] % 4 == 0 and p.is_contiguous()]
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.flat = torch.empty(off, device=dev, dtype=torch.float32) if self.params else None
        self.entries = (L.WtEntry * max(1, len(self.params)))()
        self._sentinel = None
        self._side = None       # the copies are written beside the step's forward (only its backward reads them): a stream of their own

    def refresh(self) -> None:
        if not self.params:
            return
        ptrs = tuple(p.data_ptr() for p in self.params)
        if ptrs != self._sentinel:  # first call, or a parameter's storage was re-bound
            base = self.flat.data_ptr()
            for e, p, o, ptr in zip(self.entries, self.params, self.offsets, ptrs):
                e.w, e.wt, e.n_out, e.k_in = ptr, base + 4 * o, p.shape[0], p.shape[1]
            self._sentinel = ptrs
        dev = self.flat.device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        # behind everything the caller's stream holds (the update that produced these weights, the last step's products that read the old
        # copies); the consumers wait for the refresh's event inside the library
        self._side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.device(dev):
            L.check(L.load().stlt_ctx_wt_refresh(self.context.handle, self.entries, len(self.params), self._side.cuda_stream), "stlt_ctx_wt_refresh")

    def clear(self) -> None:
        if self.params:
            # the caller's stream joins the copy stream: the flat buffer may be rewritten by the next refresh only after this step's readers,
            # and torch's allocator sees both streams ordered
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
            L.load().stlt_ctx_wt_clear(self.context.handle)


def _train_wt_on() -> bool:
    return os.environ.get("STLT_TRAIN_WT", "1") != "0"


class Trainer:
    def __init__(self, model, dataset_name: str = "something", learning_rate: float = 5e-5, weight_decay: float = 1e-3,
                 clip_val: float = 5.0, warmup_steps: int = 0, total_steps: int = 1, rank: int = 0, world: int = 1,
                 fused_optimizer: Optional[bool] = None):
        self.model, self.dataset_name, self.clip_val = model, dataset_name, clip_val
        self.rank, self.world = rank, world
        first = next(iter(model.parameters()), None)
        on_gpu = first is not None and first.is_cuda
        if fused_optimizer is None:  # the fused path reads a flat gradient buffer on a GPU
            fused_optimizer = on_gpu
        self.fused = fused_optimizer
        # Stlt fills its own flat buffer in the native reverse sweep; any other model (the fusion models) gets its .grad
        # tensors bound to one
        # what this loop leaves inside the library between calls (transposed weight copies, deferred block weight gradients, the side stream
        # of its reverse sweeps) hangs off its own handle: two trainers in one process never see each other's
        self.context = ops.TrainContext() if on_gpu else None
        self.bound = BoundFlatGrads(model, self.context) if (fused_optimizer and not hasattr(model, "_grad_params")) else None
        opt = FusedAdamW if fused_optimizer else torch.optim.AdamW  # same defaults (betas 0.9/0.999, eps 1e-8)
        self.optimizer = opt(add_weight_decay(model, weight_decay), lr=learning_rate)
        self.scheduler = linear_schedule_with_warmup(self.optimizer, warmup_steps, total_steps)
        self._comm_stream = None
        self.transposed = TransposedWeights(model, self.context) if (fused_optimizer and on_gpu and _train_wt_on()) else None

    def _sync_slice(self, flat: torch.Tensor, lo: int, hi: int) -> None:
        """Called by the native backward when flat[lo:hi] is final: average it over the ranks on a side stream, so the
        all-reduce of the temporal tower's gradients overlaps the spatial half of the reverse sweep."""
        if hi <= lo:
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=flat.device)
        self._comm_stream.wait_stream(torch.cuda.current_stream(flat.device))
        with torch.cuda.stream(self._comm_stream):
            part = flat[lo:hi]
            torch.distributed.all_reduce(part, op=torch.distributed.ReduceOp.SUM)
            part.div_(self.world)

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
        """One optimisation step on this rank's shard of the global batch (train.py:119-135)."""
        if self.transposed is None:
            return self._step(batch)
        self.transposed.refresh()  # the input-gradient products of this step read transposed copies of the weights as they are now
        try:
            return self._step(batch)
        finally:
            self.transposed.clear()

    def _step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
        self.model.train(True)
        if self.bound is not None:
            return self._step_bound(batch)
        self.optimizer.zero_grad()
        self.model._flat_grads_only = self.fused
        self.model._grad_sync = self._sync_slice if (self.fused and self.world > 1) else None
        self.model._train_context = self.context  # the reverse sweep names this trainer's context (weight copies, side stream)
        logits = self.model(batch)
        if self.fused:  # loss + dlogits in one pass per head, then straight into the native reverse sweep
            loss, heads = 0.0, list(logits.values())
            grads = []
            for v in heads:
                l, g = fused_criterion(v, batch["labels"], self.dataset_name, 1.0 / len(heads))
                loss = loss + l
                grads.append(g)
            torch.autograd.backward(heads, grads)
        else:
            loss = criterion(logits, batch["labels"], self.dataset_name)
            loss.backward()
        if self.fused:
            # the reverse sweep left every gradient in one flat buffer: all-reduce it in place, then norm + clip +
            # AdamW straight from it (no per-parameter .grad tensors, no flatten / unflatten copies)
            flat, layout = self.model._last_flat_grad, self.model._flat_layout
            if self._comm_stream is not None:  # the slices were reduced on the side stream as they became final
                torch.cuda.current_stream(flat.device).wait_stream(self._comm_stream)
            grad_norm = self.optimizer.step_flat(flat, layout, self.clip_val)
        else:
            allreduce_gradients(self.model, self.world)
            grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_val)
            self.optimizer.step()
        self.scheduler.step()
        # the two hooks only apply to this trainer's own backward: a later hand-written loop on the same model gets
        # ordinary per-parameter .grad tensors again
        self.model._flat_grads_only = False
        self.model._grad_sync = None
        self.model._train_context = None
        return {"loss": loss.detach(), "grad_norm": grad_norm.detach()}

    def _step_bound(self, batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
        """The same step for a model whose gradients arrive through autograd (CAF / CACNF / LCF): .grad views of one flat
        buffer, native criterion per logit head, in-place all-reduce, fused clip + AdamW."""
        self.bound.zero()
        logits = self.model(batch)
        heads = list(logits.values())
        loss, grads = 0.0, []
        for v in heads:
            l, g = fused_criterion(v, batch["labels"], self.dataset_name, 1.0 / len(heads))
            loss = loss + l
            grads.append(g)
        self.bound.accumulating = True
        try:
            # the blocks' weight-gradient products (2 - 4 per block, 34 blocks in CACNF) are queued and run as a few grouped launches when
            # the backward pass is through (ops.deferred_block_weight_grads; STLT_BLOCK_DW_DEFER=0: per block, as before)
            if _BLOCK_DW_DEFER:
                with ops.deferred_block_weight_grads(self.context):
                    torch.autograd.backward(heads, grads)
            else:
                torch.autograd.backward(heads, grads)
        finally:
            self.bound.accumulating = False
        flat = self.bound.flat
        if self.world > 1:
            torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
            flat.div_(self.world)
        grad_norm = self.optimizer.step_flat(flat, self.bound.layout(), self.clip_val)
        self.scheduler.step()
        return {"loss": loss.detach(), "grad_norm": grad_norm.detach()}

    def fit(self, batches: Iterable[Dict[str, torch.Tensor]], device) -> List[Dict[str, float]]:
        """Run over an iterable of GLOBAL batches, each rank taking its contiguous shard."""
        log = []
        for batch in batches:
            n = batch["categories"].shape[0]
            if self.world > 1 and n % self.world != 0:
                # the loss is a batch mean and the gradients are averaged with 1/world: unequal shards would weight clips
                # unequally, and a rank without clips would leave the others waiting in the all-reduce
                raise ValueError(f"global batch of {n} clips does not divide over {self.world} ranks: drop or pad the last batch "
                                 f"(DataLoader(drop_last=True)), as DistributedSampler does")
            mine = D.shard_batch(batch, self.rank, self.world)
            mine = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in mine.items()}
            out = self.step(mine)
            if self.world > 1:  # report the global-batch mean loss
                l = out["loss"].clone()
                torch.distributed.all_reduce(l)
                out["loss"] = l / self.world
            log.append({k: float(v) for k, v in out.items()})
        return log

    def fit_epochs(self, train_batches, val_batches, evaluator, epochs: int, device, save_model_path: Optional[str] = None,
                   save_backbone_path: Optional[str] = None, on_epoch: Optional[Callable[[dict], None]] = None) -> List[dict]:
        """The epoch shell of the reference's `train()` (src/train.py:115-152): per epoch, the optimisation steps over the epoch's
        GLOBAL batches (`fit`), then `model.train(False)`, `evaluator.reset()`, the validation batches under no_grad into the
        evaluator (utils/evaluation.py: counters / score tables stay on the device, one read-back in `evaluate()`), and when
        `evaluator.is_best()` rank 0 writes `model.state_dict()` to `save_model_path` and — if asked — `model.backbone.state_dict()`
        to `save_backbone_path` (what `Stlt(config)` with `load_backbone_path` reads back, models.py:130-134,170-176).

        `train_batches`: a re-iterable of collated batches holding `labels` (a list, a DataLoader), or a callable epoch -> iterable
        (a fresh shuffle per epoch).  Every rank passes the same global batches and takes its contiguous shard; the evaluators sum
        their counters / gather their tables over the default process group.  The scheduler's horizon is the caller's:
        `Trainer(total_steps=epochs * len(train_loader), warmup_steps=warmup_epochs * len(train_loader))` as train.py:108-113.
        -> one dict per epoch: {"epoch", "steps": [{"loss", "grad_norm"}...], "metrics", "is_best", "saved"}."""
        history = []
        for epoch in range(int(epochs)):
            steps = self.fit(train_batches(epoch) if callable(train_batches) else train_batches, device)
            self.model.train(False)
            evaluator.reset()
            with torch.no_grad():
                for batch in (val_batches(epoch) if callable(val_batches) else val_batches):
                    mine = D.shard_batch(batch, self.rank, self.world)
                    if mine["categories"].shape[0] == 0:  # more ranks than clips in the last batch
                        continue
                    mine = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in mine.items()}
                    evaluator.process(self.model(mine), mine["labels"])
            metrics = evaluator.evaluate()
            best = bool(evaluator.is_best())  # every rank evaluates (the evaluators' collectives need all of them); rank 0 writes
            saved = []
            if best and self.rank == 0:
                if save_model_path:
                    torch.save(self.model.state_dict(), save_model_path)
                    saved.append(save_model_path)
                if save_backbone_path:
                    torch.save(self.model.backbone.state_dict(), save_backbone_path)
                    saved.append(save_backbone_path)
            rec = {"epoch": epoch, "steps": steps, "metrics": dict(metrics), "is_best": best, "saved": saved}
            if on_epoch is not None:
                on_epoch(rec)
            history.append(rec)
        return history
