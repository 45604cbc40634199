"""Checkpoint round trip (reference src/train.py:147-152 -> src/inference.py:59-69): tests/golden/ckpt_trained_nano.npz holds the tensors of
a checkpoint file that `Trainer.fit_epochs` WROTE on an MI355X after training (tools/make_ckpt_fixture.py) and the logits the package
computed from it there.  The oracle (pinned to the reference) and — in the build container, where /root/reference exists — the
reference's own `Stlt` with strict=True reproduce those logits from the same tensors; on a GPU the package's module does."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle import stlt_oracle as O


def _load(pkg):
    z = np.load(os.path.join(GOLDEN, "ckpt_trained_nano.npz"))
    sd = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("__")}
    val = [pkg.synth.fit_batch("val", 0, i) for i in range(pkg.synth.FIT_TASK["val_batches"])]
    return z, sd, val


def test_written_checkpoint_has_the_reference_schema_and_the_oracle_reproduces_its_logits(pkg):
    z, sd, val = _load(pkg)
    model = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("nano")))
    want = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in sd.items()} == want  # names and shapes: what load_state_dict(strict=True) checks
    assert sd["backbone.frames_embeddings.position_ids"].dtype == torch.int64
    H = pkg.synth.CONFIGS["nano"]["num_attention_heads"]
    with torch.no_grad():
        got = torch.cat([O.stlt_forward(sd, b, H)["stlt"] for b in val]).numpy()
    assert np.abs(got - z["__logits__"]).max() <= 1e-4
    # it is a TRAINED checkpoint: the weights moved away from the seeded initialisation and the saved epoch's accuracy is the recorded one
    init = pkg.synth.make_state_dict(want, seed=pkg.synth.FIT_TASK["weight_seed"])
    assert float((sd["prediction_head.fc1.weight"] - init["prediction_head.fc1.weight"]).abs().max()) > 1e-3
    labels = torch.cat([b["labels"] for b in val]).numpy()
    assert float((got.argmax(1) == labels).mean()) == float(z["__top1__"][z["__saved_epochs__"]][-1])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree exists in the build container only")
def test_written_checkpoint_loads_strictly_into_the_reference_model():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_ckpt_roundtrip", os.path.join(ROOT, "tools", "check_ckpt_roundtrip.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    saved_path, saved_mods = list(__import__("sys").path), set(__import__("sys").modules)
    try:
        r = mod.check()
    finally:  # the reference's `modelling` / `utils` packages must not leak into the other tests of this process
        import sys
        sys.path[:] = saved_path
        for m in set(sys.modules) - saved_mods:
            if m.split(".")[0] in ("modelling", "utils"):
                del sys.modules[m]
    assert r["ok"] and r["max_abs_logit_diff"] <= 1e-4, r


@pytest.mark.gpu
def test_written_checkpoint_round_trips_through_the_package_module(pkg):
    z, sd, val = _load(pkg)
    model = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("nano")))
    model.load_state_dict(sd, strict=True)
    model.train(False).to("cuda")
    with torch.no_grad():
        got = torch.cat([model({k: v.to("cuda") for k, v in b.items()})["stlt"] for b in val]).cpu().numpy()
    assert np.abs(got - z["__logits__"]).max() <= 1e-5
