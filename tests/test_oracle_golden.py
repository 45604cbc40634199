"""Pin the CPU oracle to the reference's own outputs (tests/golden/, made by tools/gen_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import golden_case
from oracle import stlt_oracle as O


@pytest.mark.parametrize("name", ["micro", "cfg1", "cfg2", "cfg2p", "refdef", "heads", "odd", "cfg4"])
def test_oracle_logits_match_reference(name, synth):
    sd, batch, z, meta = golden_case(name)
    H = synth.CONFIGS[name]["num_attention_heads"]
    with torch.no_grad():
        got = O.stlt_forward(sd, batch, H)["stlt"].numpy()
    assert got.shape == z["logits"].shape
    # fp32 oracle vs fp32 reference: same math, possibly different summation order
    assert np.abs(got - z["logits"]).max() <= 2e-5
    # and both sit on the fp64 reference run
    assert np.abs(got.astype(np.float64) - z["logits_fp64"]).max() <= 2e-5


@pytest.mark.parametrize("name", ["micro", "cfg1"])
def test_oracle_fp64_and_taps(name, synth):
    sd, batch, z, meta = golden_case(name)
    H = synth.CONFIGS[name]["num_attention_heads"]
    taps = {}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    with torch.no_grad():
        got = O.stlt_forward(sd, b64, H, dtype=torch.float64, taps=taps)["stlt"].numpy()
    assert np.abs(got - z["logits_fp64"]).max() <= 1e-9  # restatement == reference in fp64
    for k in z.files:
        if not k.startswith("tap_"):
            continue
        ref = z[k]
        mine = taps[k[4:]].numpy()
        assert mine.shape == ref.shape, k
        if k.startswith("tap_spatial"):
            # padded object rows are never read downstream; compare real tokens only
            keep = ~batch["src_key_padding_mask_boxes"].numpy()
            assert np.abs(mine[keep] - ref[keep]).max() <= 5e-5, k
        else:
            assert np.abs(mine - ref).max() <= 5e-5, k


def test_oracle_backbone_output_time_major(synth):
    sd, batch, z, meta = golden_case("cfg1")
    H = synth.CONFIGS["cfg1"]["num_attention_heads"]
    with torch.no_grad():
        out = O.backbone_forward(sd, batch, H, prefix="backbone.")
    # reference returns (T,B,d); every row incl. padded frames is defined
    assert np.abs(out.transpose(0, 1).numpy() - z["backbone_tbd"]).max() <= 5e-5


def test_dropout_mask_definition_statistics():
    """The build's own counter-based dropout mask (oracle restatement of csrc/common.h: stlt_keep): keep rate, independence of
    neighbouring elements and of sites / seeds, and indices beyond 32 bits folding in instead of wrapping."""
    n = 1 << 20
    idx = np.arange(n, dtype=np.uint64)
    for p in (0.1, 0.5):
        k = O.dropout_keep(p, 1234, 8, idx)
        assert abs(k.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n)
        # neighbours, sites and seeds are uncorrelated: joint keep rate = product of the marginals
        for other in (np.roll(k, 1), O.dropout_keep(p, 1234, 9, idx), O.dropout_keep(p, 1235, 8, idx)):
            assert abs((k & other).mean() - (1 - p) ** 2) < 6 * np.sqrt((1 - p) ** 2 / n)
    hi = O.dropout_keep(0.5, 7, 3, idx + (np.uint64(1) << np.uint64(32)))
    assert abs((hi == O.dropout_keep(0.5, 7, 3, idx)).mean() - 0.5) < 0.01  # the high word changes the stream
    assert O.dropout_keep(0.0, 1, 1, idx[:1000]).all()
