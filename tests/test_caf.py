"""CAF / CACNF on precomputed appearance features (BASELINE config 5) against fixtures captured
from the reference's own modules (tools/gen_golden_caf.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import caf_oracle as CO

NAME = "cfg1"
EXTRA = dict(appearance_num_frames=32, num_appearance_layers=2, num_fusion_layers=2)


def _case(synth, model_name):
    z = np.load(os.path.join(GOLDEN, f"{model_name}_cfg1.npz"))
    meta = json.load(open(os.path.join(GOLDEN, f"{model_name}_cfg1_schema.json")))
    c = synth.CONFIGS[NAME]
    sd = synth.make_state_dict({k: tuple(v) for k, v in meta["keys"].items()}, seed=meta["weight_seed"])
    batch = synth.make_batch(meta["batch"], c["T"], c["N"], seed=meta["input_seed"])
    batch["appearance_features"] = synth.make_appearance_features(meta["batch"], seed=meta["feature_seed"])
    return z, meta, sd, batch, c


@pytest.mark.parametrize("model_name", ["caf", "cacnf"])
def test_caf_oracle_matches_reference(synth, model_name):
    z, meta, sd, batch, c = _case(synth, model_name)
    fwd = CO.caf_forward if model_name == "caf" else CO.cacnf_forward
    with torch.no_grad():
        out = fwd(sd, batch, c["num_attention_heads"])
    assert set(out) == set(z.files)
    for k in z.files:
        assert np.abs(out[k].numpy() - z[k]).max() <= 3e-5, k


@pytest.mark.parametrize("model_name", ["caf", "cacnf"])
def test_caf_state_dict_keys_are_the_reference_non_r3d_keys(pkg, model_name):
    _, meta, _, _, _ = _case(pkg.synth, model_name)
    cls = pkg.models_factory[model_name]
    m = cls(pkg.MultimodalModelConfig(**dict(pkg.synth.model_kwargs(NAME), **EXTRA)))
    sd = m.state_dict()
    assert list(sd) == list(meta["keys"])
    assert all(list(sd[k].shape) == meta["keys"][k] for k in sd)


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["caf", "cacnf"])
def test_caf_gpu_matches_reference(pkg, model_name):
    z, meta, sd, batch, c = _case(pkg.synth, model_name)
    cls = pkg.models_factory[model_name]
    m = cls(pkg.MultimodalModelConfig(**dict(pkg.synth.model_kwargs(NAME), **EXTRA)))
    m.load_state_dict(sd, strict=True)
    m.train(False).to("cuda")
    out = m({k: v.to("cuda") for k, v in batch.items()})
    assert tuple(out) == m.logit_names or set(out) == set(m.logit_names)
    for k in z.files:
        got = out[k].cpu().numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - z[k]).max() <= 1e-4, k


@pytest.mark.gpu
def test_caf_gpu_full_width_matches_oracle(pkg):
    """d=768 / 12 heads / T=32, N=7 (the cfg2 layout shapes of BASELINE config 5), default 4+4 fusion/appearance layers."""
    c = pkg.synth.CONFIGS["cfg2"]
    kw = dict(pkg.synth.model_kwargs("cfg2"), appearance_num_frames=32)
    m = pkg.CrossAttentionCentralNetFusion(pkg.MultimodalModelConfig(**kw))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    m.load_state_dict(sd)
    m.train(False).to("cuda")
    batch = pkg.synth.make_batch(2, c["T"], c["N"], seed=8)
    batch["appearance_features"] = pkg.synth.make_appearance_features(2, seed=9)
    out = m({k: v.to("cuda") for k, v in batch.items()})
    with torch.no_grad():
        ref = CO.cacnf_forward(sd, batch, c["num_attention_heads"])
    for k in ref:
        assert (out[k].cpu() - ref[k]).abs().max().item() <= 1e-4, k
