#!/usr/bin/env python3
"""Throughput of CACNF / CAF on precomputed appearance features (BASELINE config 5): cfg2 layout shapes, (B,2048,2,4,4)
feature maps, 4 appearance + 4 fusion layers.  One JSON line on rank 0; batch sharded over ranks.

    python tools/bench_caf.py [--batch 512] [--skip-padding]          # inference: one native call per step (no_grad)
    python tools/bench_caf.py --train [--batch 32] [--dropout 0.1]    # optimisation step: forward + loss + backward + AdamW
"""
import argparse, importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="cacnf", choices=["caf", "cacnf"])
    ap.add_argument("--skip-padding", action="store_true", help="layout branch on the real tokens / frames only (inference)")
    ap.add_argument("--train", action="store_true", help="time an optimisation step instead of the inference call")
    ap.add_argument("--stock-loop", action="store_true", help="--train: zero_grad / F.cross_entropy / clip_grad_norm_ / torch.optim.AdamW instead of train.Trainer")
    ap.add_argument("--dropout", type=float, default=0.1, help="--train: hidden_dropout_prob (reference default 0.1)")
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
    rank, world = pkg.dist.init_distributed()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    c = pkg.synth.CONFIGS["cfg2"]
    kw = dict(pkg.synth.model_kwargs("cfg2"), appearance_num_frames=32)
    if args.train:
        kw["hidden_dropout_prob"] = args.dropout
    m = pkg.models_factory[args.model](pkg.MultimodalModelConfig(**kw))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234)
    m.load_state_dict(sd)
    m.train(args.train).to(dev)
    B = args.batch
    batch = pkg.synth.make_batch(B, c["T"], c["N"], seed=1 + rank)
    batch["appearance_features"] = pkg.synth.make_appearance_features(B, seed=rank)
    batch = {k: v.to(dev) for k, v in batch.items()}
    if args.skip_padding:
        for mod in m.modules():
            if isinstance(mod, pkg.StltBackbone):
                mod.skip_padding = True
    if args.train:
        labels = torch.randint(0, c["num_classes"], (B,), generator=torch.Generator().manual_seed(rank)).to(dev)
        batch["labels"] = labels
        if args.stock_loop:  # the reference's loop with stock torch ops: zero_grad, CE per head, backward, clip, AdamW
            opt = torch.optim.AdamW(pkg.train.add_weight_decay(m, 1e-3), lr=5e-5)

            def step():
                opt.zero_grad(set_to_none=True)
                out = m(batch)
                loss = sum(torch.nn.functional.cross_entropy(v, labels) for v in out.values()) / len(out)
                loss.backward()
                if world > 1:
                    pkg.train.allreduce_gradients(m, world)
                torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
                opt.step()
                return out
        else:  # train.Trainer: gradients bound to one flat buffer, native criterion, fused clip + AdamW
            tr = pkg.train.Trainer(m, "something", learning_rate=5e-5, weight_decay=1e-3, clip_val=5.0, warmup_steps=0, total_steps=100000,
                                   rank=rank, world=world)

            def step():
                res = tr.step(batch)
                return {"loss": res["loss"].reshape(1)}
    else:
        def step():
            with torch.no_grad():  # with grad enabled the module takes its autograd path
                return m(batch)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        what = "train step (forward + CE over the heads + backward + clip + AdamW)" if args.train else "forward"
        print(json.dumps({"metric": f"clips/s {args.model.upper()} {what} on precomputed appearance features", "value": round(world * B / dt, 1),
                          "n_gpus": world, "ms_per_step": round(dt * 1e3, 2), "per_gpu_batch": B, "dtype": "f32", "skip_padding": bool(args.skip_padding),
                          "finite": bool(all(torch.isfinite(v).all() for v in out.values()))}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
