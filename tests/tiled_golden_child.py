"""Child process of test_model_gpu.py::test_bench_sized_launches_reproduce_the_golden_rows: the golden batch `name` tiled `reps` times
through Stlt.forward with whatever library switches the parent put in the environment (the library reads them once per process);
exits non-zero unless every replica's logits are within 1e-4 of its golden row."""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import PKG_NAME, golden_case  # noqa: E402


def main():
    name, reps = sys.argv[1], int(sys.argv[2])
    pkg = importlib.import_module(PKG_NAME)
    sd, batch, z, meta = golden_case(name)
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
    m.load_state_dict(sd, strict=True)
    m = m.train(False).to("cuda")
    if os.environ.get("STLT_FUSED_MHSA") == "0":
        assert not pkg._lib.load().stlt_fused_mhsa_active(32, 768, 12)
    big = {k: v.repeat(reps, *([1] * (v.dim() - 1))).to("cuda") for k, v in batch.items()}
    with torch.no_grad():
        out = m(big)["stlt"].cpu()
    err = float((out - torch.from_numpy(z["logits"]).repeat(reps, 1)).abs().max())
    print("max |logit - golden row| =", err)
    assert torch.isfinite(out).all() and err <= 1e-4, err


if __name__ == "__main__":
    main()
