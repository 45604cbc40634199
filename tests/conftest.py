import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG_NAME = "revisiting-spatial-temporal-layouts_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_collection_modifyitems(config, items):
    # gpu tests are selected with -m gpu; if somebody runs the whole suite on a CPU box, skip them.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG_NAME + ".synth")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    with open(os.path.join(GOLDEN, f"{name}_schema.json")) as f:
        meta = json.load(f)
    return z, meta


def golden_case(name):
    """(state_dict, batch, golden npz, meta) rebuilt from seeds + committed expected outputs."""
    synth = importlib.import_module(PKG_NAME + ".synth")
    z, meta = load_golden(name)
    c = synth.CONFIGS[name]
    shapes = {k: tuple(v["shape"]) for k, v in meta["keys"].items()}
    sd = synth.make_state_dict(shapes, seed=meta["weight_seed"])
    batch = synth.make_batch(meta["batch"], c["T"], c["N"], dataset=c["dataset"], seed=meta["input_seed"])
    assert int(batch["categories"].sum()) == int(z["digest_categories"][0])
    assert abs(float(batch["boxes"].double().sum()) - float(z["digest_boxes"][0])) < 1e-9
    assert np.array_equal(batch["lengths"].numpy(), z["lengths"])
    return sd, batch, z, meta
