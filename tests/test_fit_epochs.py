"""The epoch shell of `train()` (reference src/train.py:115-152) — per epoch: optimisation steps, validation through the evaluator,
`is_best()` -> rank-0 save — against tests/golden/fit_micro.npz, captured by tools/gen_golden_fit.py from the reference's own
`Stlt` + `Criterion` + `add_weight_decay` + AdamW + scheduler + `EvaluatorSomething` driven through the reference loop's statements.

CPU test: the shell around the oracle under torch autograd (stock optimiser).  GPU test: the native step, the device evaluator, and
the two files the shell writes, loaded back strictly into fresh modules."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from test_train_gloo import _OracleStlt

WATCH = ["prediction_head.fc2.bias", "prediction_head.fc1.weight", "backbone.frames_embeddings.position_embeddings.weight",
         "backbone.frames_embeddings.layout_embedding.transformer.layers.0.self_attn.in_proj_weight", "backbone.transformer.layers.0.linear2.weight"]


def _run(pkg, model, device, tmp_path, backbone_file):
    task = pkg.synth.FIT_TASK
    c = pkg.synth.CONFIGS[task["config"]]
    nb = task["train_batches"]
    tr = pkg.train.Trainer(model, "something", learning_rate=task["lr"], weight_decay=task["weight_decay"], clip_val=task["clip_val"],
                           warmup_steps=task["warmup_epochs"] * nb, total_steps=task["epochs"] * nb)
    val = [pkg.synth.fit_batch("val", 0, i) for i in range(task["val_batches"])]
    ev = pkg.evaluators_factory["something"](sum(b["labels"].shape[0] for b in val), c["num_classes"], ("stlt",))
    model_file = str(tmp_path / "model.pt")
    seen = []

    def on_epoch(rec):  # which epoch's weights are in the file right now
        seen.append(os.path.getmtime(model_file) if os.path.exists(model_file) else None)

    hist = tr.fit_epochs(lambda e: [pkg.synth.fit_batch("train", e, i) for i in range(nb)], val, ev, task["epochs"], device,
                         save_model_path=model_file, save_backbone_path=backbone_file, on_epoch=on_epoch)
    return hist, model_file


def _check_history(hist, z, loss_tol):
    assert len(hist) == int(z["epochs"][0])
    assert float(z["min_margin"][0]) > 1e-2  # every top-1 / top-5 decision of the fixture is far from a rounding flip
    for e, rec in enumerate(hist):
        assert rec["epoch"] == e and rec["is_best"] == bool(z["saved"][e]), (e, rec["is_best"])
        assert bool(rec["saved"]) == bool(z["saved"][e])
        got = (rec["metrics"]["stlt_top1_accuracy"], rec["metrics"]["stlt_top5_accuracy"])
        assert got == tuple(z[f"metrics{e}"]), (e, got, z[f"metrics{e}"])
        mean_loss = float(np.mean([s["loss"] for s in rec["steps"]]))
        assert abs(mean_loss - float(z[f"mean_loss{e}"][0])) <= loss_tol, (e, mean_loss, float(z[f"mean_loss{e}"][0]))


def test_epoch_shell_on_the_cpu_oracle_matches_the_reference_loop(pkg, tmp_path):
    z = np.load(os.path.join(GOLDEN, "fit_micro.npz"))
    task = pkg.synth.FIT_TASK
    c = pkg.synth.CONFIGS[task["config"]]
    shapes = {k: tuple(v.shape) for k, v in pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(task["config"]))).state_dict().items()}
    sd = pkg.synth.make_state_dict(shapes, seed=task["weight_seed"])
    model = _OracleStlt(sd, c["num_attention_heads"])
    torch.set_num_threads(4)
    hist, model_file = _run(pkg, model, "cpu", tmp_path, None)
    _check_history(hist, z, 2e-4)
    # the file holds the weights of the LAST best epoch (here the last epoch), not of the first
    saved = torch.load(model_file)
    names = model.names
    for i, k in enumerate(WATCH):
        got = saved[f"ps.{names.index(k)}"].reshape(-1)[:64].numpy()
        assert np.abs(got - z[f"saved_p{i}"]).max() <= 2e-3, k  # lr 6e-3 x 12 Adam steps amplify rounding


@pytest.mark.gpu
def test_epoch_shell_native_matches_the_reference_loop_and_its_files_load_strictly(pkg, tmp_path):
    z = np.load(os.path.join(GOLDEN, "fit_micro.npz"))
    task = pkg.synth.FIT_TASK
    kw = pkg.synth.model_kwargs(task["config"])
    model = pkg.Stlt(pkg.StltModelConfig(**kw))
    model.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=task["weight_seed"]))
    model.to("cuda")
    backbone_file = str(tmp_path / "backbone.pt")
    hist, model_file = _run(pkg, model, "cuda", tmp_path, backbone_file)
    _check_history(hist, z, 5e-4)
    saved, saved_bb = torch.load(model_file, map_location="cpu"), torch.load(backbone_file, map_location="cpu")
    assert len(saved) == int(z["n_model_keys"][0]) == len(model.state_dict()) and len(saved_bb) == int(z["n_backbone_keys"][0]) == len(saved) - 6
    for i, k in enumerate(WATCH):
        assert np.abs(saved[k].reshape(-1)[:64].numpy() - z[f"saved_p{i}"]).max() <= 2e-3, k
    assert np.abs(saved_bb[WATCH[2].replace("backbone.", "", 1)].reshape(-1)[:64].numpy() - z["saved_backbone_p0"]).max() <= 2e-3
    # round trip inside the package: strict load of both files (the reference reads them with Stlt.load_state_dict, inference.py:59-69, and
    # StltBackbone.from_pretrained, models.py:130-134), and the loaded model reproduces the saved model's validation logits
    fresh = pkg.Stlt(pkg.StltModelConfig(**kw))
    fresh.load_state_dict(saved, strict=True)
    fresh.train(False).to("cuda")
    val = [pkg.synth.fit_batch("val", 0, i) for i in range(task["val_batches"])]
    with torch.no_grad():
        got = torch.cat([fresh({k: v.to("cuda") for k, v in b.items()})["stlt"] for b in val]).cpu().numpy()
    assert np.abs(got - z["saved_val_logits"]).max() <= 2e-2  # the reference's logits after ITS 12 steps: trajectories differ by rounding x Adam
    bb = pkg.StltBackbone.from_pretrained(pkg.StltModelConfig(**dict(kw, load_backbone_path=backbone_file)))
    assert all(torch.equal(v, saved_bb[k]) for k, v in bb.state_dict().items())
