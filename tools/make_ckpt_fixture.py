#!/usr/bin/env python3
"""Runs ON THE GPU BOX: train the `nano` model on the fit task (synth.FIT_TASK) with `Trainer.fit_epochs` — native forward, reverse
sweep, fused clip + AdamW, device evaluator, best-epoch saves — then write what was SAVED (the checkpoint file's tensors as numeric
arrays, keyed by their state-dict names) plus the logits the saved model gives on the seeded validation batches:

    gpurun -- python tools/make_ckpt_fixture.py --out gpurun_out/r6/ckpt_trained_nano.npz
    cp gpurun_out/r6/ckpt_trained_nano.npz tests/golden/

tools/check_ckpt_roundtrip.py (build container) loads that file into the REFERENCE's `Stlt` with strict=True and reproduces the
logits; tests/test_ckpt_roundtrip.py does the same with the oracle (anywhere) and with this package's modules (GPU)."""
import argparse, importlib, os, sys, tempfile
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="nano")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    task, synth = pkg.synth.FIT_TASK, pkg.synth
    c = synth.CONFIGS[args.config]
    model = pkg.Stlt(pkg.StltModelConfig(**synth.model_kwargs(args.config)))
    model.load_state_dict(synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=task["weight_seed"]))
    model.to("cuda")
    nb = task["train_batches"]
    tr = pkg.train.Trainer(model, "something", learning_rate=task["lr"], weight_decay=task["weight_decay"], clip_val=task["clip_val"],
                           warmup_steps=task["warmup_epochs"] * nb, total_steps=task["epochs"] * nb)
    val = [synth.fit_batch("val", 0, i) for i in range(task["val_batches"])]
    ev = pkg.evaluators_factory["something"](sum(b["labels"].shape[0] for b in val), c["num_classes"], model.logit_names)
    with tempfile.TemporaryDirectory() as d:
        mf, bf = os.path.join(d, "model.pt"), os.path.join(d, "backbone.pt")
        hist = tr.fit_epochs(lambda e: [synth.fit_batch("train", e, i) for i in range(nb)], val, ev, task["epochs"], "cuda", mf, bf)
        saved, saved_bb = torch.load(mf, map_location="cpu"), torch.load(bf, map_location="cpu")
    fresh = pkg.Stlt(pkg.StltModelConfig(**synth.model_kwargs(args.config)))
    fresh.load_state_dict(saved, strict=True)
    fresh.train(False).to("cuda")
    with torch.no_grad():
        logits = torch.cat([fresh({k: v.to("cuda") for k, v in b.items()})["stlt"] for b in val]).cpu().numpy()
    out = {k: v.numpy() for k, v in saved.items()}
    assert all(torch.equal(v, saved["backbone." + k]) for k, v in saved_bb.items()) and len(saved_bb) == len(saved) - 6
    out["__logits__"] = logits
    out["__top1__"] = np.array([r["metrics"]["stlt_top1_accuracy"] for r in hist])
    out["__saved_epochs__"] = np.array([r["is_best"] for r in hist])
    out["__mean_loss__"] = np.array([float(np.mean([s["loss"] for s in r["steps"]])) for r in hist])
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print(f"[ckpt] {args.config}: {len(saved)} keys, {sum(v.numel() for v in saved.values())} elements, top1 per epoch {out['__top1__']}, saved {out['__saved_epochs__']}, "
          f"file {os.path.getsize(args.out)} bytes")


if __name__ == "__main__":
    main()
