#!/usr/bin/env python3
"""Training-step throughput of the STLT path (BASELINE.json config 3: cfg2 shapes, 64 clips per GPU, gradients
all-reduced over RCCL when launched under torch.distributed.run).  Prints one JSON line on rank 0."""
import argparse, importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--skip-padding", action="store_true", help="run forward and backward on the real tokens / frames only")
    ap.add_argument("--dropout", type=float, default=0.0)
    ap.add_argument("--stock-optimizer", action="store_true", help="torch clip_grad_norm_ + AdamW instead of the fused kernels")
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
    one_gpu = os.environ.get("STLT_BENCH_ONE_GPU") == "1"  # plumbing test of the multi-rank path on a single-GPU box (gloo, all ranks on cuda:0)
    rank, world = pkg.dist.init_distributed("gloo" if one_gpu else None)
    dev = torch.device("cuda", 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    c = pkg.synth.CONFIGS[args.config]
    kw = pkg.synth.model_kwargs(args.config)
    kw["hidden_dropout_prob"] = args.dropout
    model = pkg.Stlt(pkg.StltModelConfig(**kw))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    model.load_state_dict(sd)
    model.to(dev)
    model.backbone.skip_padding = args.skip_padding
    tr = pkg.train.Trainer(model, "something", warmup_steps=2, total_steps=1000, rank=rank, world=world,
                           fused_optimizer=False if args.stock_optimizer else None)
    B = args.batch
    batch = pkg.synth.make_batch(B, c["T"], c["N"], seed=1 + rank)
    batch["labels"] = torch.randint(0, c["num_classes"], (B,))
    batch = {k: v.to(dev) for k, v in batch.items()}
    for _ in range(args.warmup):
        tr.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / args.steps
    # phase split on rank 0 (events)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    model.train(True)
    tr.optimizer.zero_grad()
    ev[0].record(); logits = model(batch); ev[1].record()
    if tr.fused:
        loss, dl = pkg.train.fused_criterion(logits["stlt"], batch["labels"]); logits["stlt"].backward(dl)
    else:
        loss = pkg.train.criterion(logits, batch["labels"]); loss.backward()
    ev[2].record()
    if tr.fused:
        flat = model._last_flat_grad
        if world > 1:
            torch.distributed.all_reduce(flat); flat.div_(world)
        ev[3].record()
        tr.optimizer.step_flat(flat, model._flat_layout, 5.0); ev[4].record()
    else:
        pkg.train.allreduce_gradients(model, world); ev[3].record()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); tr.optimizer.step(); ev[4].record()
    torch.cuda.synchronize()
    ph = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    if rank == 0:
        fl = pkg.synth.flops_per_clip(c["T"], c["N"], c["hidden_size"], c["num_spatial_layers"], c["num_temporal_layers"], c["num_classes"])
        print(json.dumps({"metric": "clips/s STLT train step", "value": round(world * B / dt, 1), "n_gpus": world, "ms_per_step": round(dt * 1e3, 2),
                          "per_gpu_batch": B, "skip_padding": args.skip_padding, "dropout": args.dropout, "loss": float(out["loss"]), "tflops_fwd_bwd": round(3 * fl * B / dt / 1e12, 1),
                          "phase_ms": {"forward": round(ph[0], 2), "loss+backward": round(ph[1], 2), "grad_allreduce": round(ph[2], 2),
                                       "clip+adamw": round(ph[3], 2)}}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
