"""Collater: CPU oracle and device kernel against the reference's StltCollater outputs
(tests/golden/collate_*.npz, tools/gen_golden_collate.py).  Pure data movement: bit exact."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import collate_oracle as CO

CASES = [("something", 5, 1), ("action_genome", 9, 2)]


def _check(got, z, dataset):
    keys = ["categories", "boxes", "frame_types", "lengths", "labels", "src_key_padding_mask_boxes",
            "src_key_padding_mask_frames"] + (["scores"] if dataset == "action_genome" else [])
    assert ("scores" in got) == (dataset == "action_genome")
    for k in keys:
        a = got[k].cpu().numpy() if isinstance(got[k], torch.Tensor) else np.asarray(got[k])
        assert a.shape == z[k].shape and a.dtype == z[k].dtype, k
        assert np.array_equal(a, z[k]), k


@pytest.mark.parametrize("dataset,N,seed", CASES)
def test_collate_oracle_matches_reference(synth, dataset, N, seed):
    z = np.load(os.path.join(GOLDEN, f"collate_{dataset}.npz"))
    _check(CO.collate(synth.make_video_samples(dataset, 4, N, seed), dataset), z, dataset)


@pytest.mark.gpu
@pytest.mark.parametrize("dataset,N,seed", CASES)
def test_device_collater_matches_reference(pkg, dataset, N, seed):
    z = np.load(os.path.join(GOLDEN, f"collate_{dataset}.npz"))
    samples = pkg.synth.make_video_samples(dataset, 4, N, seed)
    got = pkg.collate.DeviceCollater(dataset, "cuda")(samples)
    _check(got, z, dataset)
    assert got["video_id"] == [s["video_id"] for s in samples]
    assert all(isinstance(v, list) or v.is_cuda for v in got.values())


@pytest.mark.gpu
def test_device_collater_feeds_the_model(pkg):
    """ragged samples -> device collater -> Stlt.forward == oracle collate -> oracle forward."""
    from oracle import stlt_oracle as O
    name = "cfg1"
    c = pkg.synth.CONFIGS[name]
    samples = pkg.synth.make_video_samples("something", 6, c["N"], 9)
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
    sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3)
    m.load_state_dict(sd)
    m.train(False).to("cuda")
    with torch.no_grad():
        got = m(pkg.collate.DeviceCollater("something", "cuda")(samples))["stlt"].cpu()
        ref = O.stlt_forward(sd, CO.collate(samples, "something"), c["num_attention_heads"])["stlt"]
    assert (got - ref).abs().max().item() <= 1e-4
