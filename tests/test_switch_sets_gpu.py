"""The library's dispatch switches are environment variables read once per process, so the default GPU run exercises the default routing only.
This file re-runs a bounded subset of the parity suite in CHILD processes under the two non-default switch sets the build's own collections
use (tools/collect_round6.sh runs the whole suite under each): every routing decision turned OFF (two-launch attention instead of the fused
kernel, no small tiles, one stream, per-launch reductions, no deferred / transposed-weight / fused-epilogue paths), and the opt-in
split-bf16 products ON.  Same tests, same tolerances: the fall-back paths are held to the bar of the default ones."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SWITCH_SETS = {
    "dispatches_off": dict(STLT_FUSED_MHSA="0", STLT_GEMM16="0", STLT_TRAIN_DW_STREAM="0", STLT_TRAIN_DEFER_REDUCE="0", STLT_ATTN16_TAIL="0",
                           STLT_BLOCK_DW_DEFER="0", STLT_FFN1_KEEP_FUSED="0", STLT_TRAIN_WT="0", STLT_ATTN_BWDX16="0", STLT_ATTN16_DROPOUT="0"),
    "split_bf16_on": dict(STLT_GEMM_SPLIT_BF16="6"),
}
# logits vs the reference goldens (every config, both schedules, skip-padding), gradients vs the fp64 oracle, the reference-captured
# optimisation steps, the epoch shell, the fusion models' goldens and gradients
SUBSET = ["tests/test_model_gpu.py", "tests/test_fit_epochs.py", "tests/test_ckpt_roundtrip.py", "tests/test_caf.py",
          "tests/test_train_gpu.py::test_gradients_match_oracle_autograd", "tests/test_train_gpu.py::test_three_training_steps_match_reference_golden"]
DESELECT = ["tests/test_model_gpu.py::test_bench_sized_launches_reproduce_the_golden_rows",  # spawns its own children; minutes, not seconds
            "tests/test_train_gpu.py::test_gradients_match_oracle_autograd[cfg4-2-True-True]",  # 10 s each: the fp64 oracle's autograd at cfg4 ...
            "tests/test_train_gpu.py::test_gradients_match_oracle_autograd[cfg4-2-True-False]",
            "tests/test_train_gpu.py::test_gradients_match_oracle_autograd[cfg2-2-False-True]",  # ... 3 s each at cfg2 (cfg1's four cases stay)
            "tests/test_train_gpu.py::test_gradients_match_oracle_autograd[cfg2-2-False-False]",
            "tests/test_model_gpu.py::test_matches_oracle_on_fresh_seed_with_scores_absent_and_present"]


# the split-bf16 switch only touches the forward products: the golden-logit tests of every config are its subset
SUBSET_OF = {"dispatches_off": SUBSET, "split_bf16_on": ["tests/test_model_gpu.py", "tests/test_caf.py", "-k", "golden"]}
MIN_PASSED = {"dispatches_off": 35, "split_bf16_on": 12}


@pytest.mark.parametrize("name", sorted(SWITCH_SETS))
def test_parity_subset_under_a_non_default_switch_set(name):
    env = dict(os.environ, **SWITCH_SETS[name])
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *SUBSET_OF[name]]
    for d in DESELECT:
        cmd += ["--deselect", d]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    last = [l for l in r.stdout.splitlines() if " passed" in l][-1]
    assert " failed" not in last and int(last.split(" passed")[0].split()[-1]) >= MIN_PASSED[name], last
    print(f"[{name}] {last}")
