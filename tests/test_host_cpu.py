"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, the drop-in
modules carry the reference's state-dict schema, and the product path refuses to run without a GPU."""
import copy
import ctypes
import json
import os
import re

import pytest
import torch

import importlib

from conftest import GOLDEN, PKG_NAME, ROOT, load_golden


def test_library_exports_every_declared_symbol(pkg):
    header = open(os.path.join(ROOT, "include", "stlt_hip.h")).read()
    declared = set(re.findall(r"\b(stlt_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = pkg._lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/stlt_hip.h but not exported"
    assert declared == set(pkg._lib.SIGNATURES), "ctypes signature table out of sync with the header"
    # ... and nothing else: the dynamic symbol table is the C-ABI, no internal C++ launcher or template instantiation leaks out
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", pkg._lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
    assert exported == declared, f"exported but not declared: {sorted(exported - declared)}; declared but not exported: {sorted(declared - exported)}"
    assert lib.stlt_version() == 110
    # workspace sizing is pure host arithmetic: callable without a GPU
    a = pkg.ops.workspace_bytes(8, 32, 7, 768, 174)
    b = pkg.ops.workspace_bytes(16, 32, 7, 768, 174)
    assert 0 < a < b
    assert pkg.ops.workspace_bytes(0, 32, 7, 768, 174) == 0


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg4"])
def test_state_dict_schema_matches_reference(pkg, name):
    _, meta = load_golden(name)
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
    sd = m.state_dict()
    assert list(sd) == list(meta["keys"])  # same keys, same order
    for k, v in sd.items():
        assert list(v.shape) == meta["keys"][k]["shape"], k
        assert str(v.dtype).replace("torch.", "") == meta["keys"][k]["dtype"], k
    assert len(m.backbone.state_dict()) == len(sd) - 6
    assert m.logit_names == ("stlt",)
    # 1-D / bias parameters are what add_weight_decay exempts (train_inference_utils.py:47); names must agree
    params = list(m.named_parameters())
    assert len(params) == 173  # 174 keys minus the position_ids buffer
    assert sum(1 for n, p in params if p.dim() == 1 or n.endswith(".bias")) == 114  # probed on the reference
    # spatial layers start as copies of the (dead) encoder_layer, like nn.TransformerEncoder's deep copies
    le = m.backbone.frames_embeddings.layout_embedding
    assert torch.equal(le.encoder_layer.linear1.weight, le.transformer.layers[2].linear1.weight)
    copy.deepcopy(m)


def test_strict_and_nonstrict_loading(pkg):
    """inference.py:59-69: strict load, falling back to strict=False when score_embeddings are absent."""
    _, meta = load_golden("cfg1")
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("cfg1")))
    sd = pkg.synth.make_state_dict({k: tuple(v["shape"]) for k, v in meta["keys"].items()})
    m.load_state_dict(sd, strict=True)
    partial = {k: v for k, v in sd.items() if "score_embeddings" not in k}
    with pytest.raises(RuntimeError):
        m.load_state_dict(partial, strict=True)
    res = m.load_state_dict(partial, strict=False)
    assert all("score_embeddings" in k for k in res.missing_keys) and not res.unexpected_keys


def test_config_surface(pkg):
    with pytest.raises(AssertionError):
        pkg.StltModelConfig(unique_categories=4)
    with pytest.raises(AssertionError):
        pkg.StltModelConfig(num_classes=174)
    c = pkg.StltModelConfig(num_classes=174, unique_categories=4)
    assert (c.hidden_size, c.num_attention_heads, c.num_spatial_layers, c.num_temporal_layers) == (768, 12, 4, 8)
    assert (c.hidden_dropout_prob, c.layer_norm_eps, c.layout_num_frames) == (0.1, 1e-12, 256)
    assert c.load_backbone_path is None and c.freeze_backbone is False
    assert pkg.models_factory["stlt"] is pkg.Stlt and pkg.model_configs_factory["stlt"] is pkg.StltModelConfig


def test_launch_time_dispatch_is_host_arithmetic(pkg):
    """The two launch-time choices of round 4 are pure host arithmetic over (shape, CU count) and can be pinned without a GPU (256 CUs
    are assumed when no device answers): which products go to the small-tile kernel (csrc/gemm16.hip) and which layers run the fused
    in-projection + attention kernel (csrc/mhsa.hip) instead of the two launches."""
    if os.environ.get("STLT_GEMM16") == "0" or os.environ.get("STLT_FUSED_MHSA") == "0":
        pytest.skip("dispatch switched off in the environment")
    lib = pkg._lib.load()
    # 64-clip batches of cfg2 (M = 2048): whole small tiles of 128 x 48 / 144 / 192 = 256 tiles; bench-sized launches stay on 256 x 128.
    # The tile is returned as columns | rows << 16 (128 rows: plain columns)
    tile = lambda rows, cols: cols if rows == 128 else (cols | (rows << 16))
    assert lib.stlt_linear_small_choice(2048, 768, 768) == 48
    assert lib.stlt_linear_small_choice(2048, 2304, 768) == 144
    assert lib.stlt_linear_small_choice(2048, 3072, 768) == 192
    assert lib.stlt_linear_small_choice(229376, 2304, 768) == 0 and lib.stlt_linear_small_choice(32768, 768, 3072) == 0
    assert lib.stlt_linear_small_choice(2048, 174, 768) == 0          # N % 4 != 0: not the kernel's shape
    # round 5, tile heights of 64 / 32 rows: the reference's default layout at its default batch (17 frames x 5 slots x 64 clips:
    # M = 5440 spatial rows = 85 x 64 -> 64 x 256 tiles = 255 / 765 / 1020 for N = 768 / 2304 / 3072; M = 1088 temporal rows = 17 x 64)
    assert lib.stlt_linear_small_choice(5440, 768, 768) == tile(64, 256) and lib.stlt_linear_small_choice(5440, 2304, 768) == tile(64, 256)
    assert lib.stlt_linear_small_choice(5440, 3072, 768) == tile(64, 256) and lib.stlt_linear_small_choice(5440, 768, 3072) == tile(64, 256)
    assert (lib.stlt_linear_small_choice(1088, 768, 768) >> 16) in (32, 64) and lib.stlt_linear_small_choice(1088, 2304, 768) == tile(64, 160)
    assert lib.stlt_linear_small_choice(14336, 768, 768) == tile(64, 96)   # 1.31 rounds of large tiles (stream-K + fix-up) against 7 whole rounds
    # input gradients (weight read as it lies: the [k][n] gather costs 5 - 15 % per k-step): 2048-row products go to the small tiles
    assert lib.stlt_input_grad_small_choice(2048, 768, 768) == 48 and lib.stlt_input_grad_small_choice(2048, 3072, 768) == 48
    assert lib.stlt_input_grad_small_choice(14336, 768, 3072) == 0
    # fused MHSA: from ~256 clips on for 32 / 64 frames (17 frames: from 1024), never for 33 frames (99 of an item's 128 rows) or 36
    # objects; 64-clip launches go to the pair (small-tile in-projection + attention core)
    used = lambda S, L, causal: int(lib.stlt_fused_mhsa_used(S, L, 768, 12, causal))
    assert [used(S, 32, 1) for S in (64, 256, 1024)] == [0, 1, 1]
    assert [used(S, 64, 1) for S in (64, 256, 1024)] == [0, 1, 1]
    assert used(1024, 17, 1) == 1 and used(64, 17, 1) == 0
    assert [used(S, 33, 1) for S in (64, 256, 1024)] == [0, 0, 0]
    assert used(32768, 7, 0) == 1 and used(32768, 8, 0) == 1 and used(32768, 36, 0) == 0
    assert lib.stlt_fused_mhsa_active(32, 768, 12) == 1 and lib.stlt_fused_mhsa_active(17, 768, 12) == 1 and lib.stlt_fused_mhsa_active(64, 768, 12) == 1
    assert lib.stlt_fused_mhsa_active(33, 768, 12) == 0
    assert lib.stlt_fused_mhsa_active(65, 768, 12) == 0 and lib.stlt_fused_mhsa_active(32, 768, 8) == 0   # > 64 tokens / head dim != 64: not taken
    # the routing switch
    assert lib.stlt_set_gemm_small_tiles(0) == 0 and lib.stlt_linear_small_choice(2048, 768, 768) == 0
    assert lib.stlt_set_gemm_small_tiles(-1) == 0 and lib.stlt_linear_small_choice(2048, 768, 768) == 48
    assert lib.stlt_set_gemm_small_tiles(7) != 0
    assert lib.stlt_set_gemm_small_tiles(-2) == 0  # back to the environment's / default setting
    # training scratch: sized for the second operand sets, refused above the category limit
    assert lib.stlt_train_scratch_bytes(64, 32, 7, 768, 4) > 0 and lib.stlt_train_scratch_bytes(64, 32, 7, 768, 129) == 0


def test_no_cpu_fallback(pkg):
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("cfg1")))
    m.train(False)
    with pytest.raises(pkg.StltHipError):
        m(pkg.synth.make_batch(2, 16, 4))


def test_frozen_backbone_stays_in_eval(pkg, tmp_path):
    """Stlt.train override (models.py:180-183) + from_pretrained (models.py:130-134)."""
    kw = pkg.synth.model_kwargs("cfg1")
    bb = pkg.StltBackbone(pkg.StltModelConfig(**kw))
    path = str(tmp_path / "bb.pt")
    torch.save(bb.state_dict(), path)
    m = pkg.Stlt(pkg.StltModelConfig(**kw, load_backbone_path=path, freeze_backbone=True))
    assert all(not p.requires_grad for p in m.backbone.parameters())
    assert all(p.requires_grad for p in m.prediction_head.parameters())
    m.train(True)
    assert m.training and not m.backbone.training
    assert torch.equal(m.backbone.transformer.layers[3].linear2.weight, bb.transformer.layers[3].linear2.weight)


def test_causal_mask_helper(pkg):
    mk = pkg.generate_square_subsequent_mask(5)
    assert mk.dtype == torch.bool and mk[0, 1] and not mk[1, 1] and not mk[3, 0]


def test_synth_invariants(pkg):
    for ds, T, N in (("something", 32, 7), ("action_genome", 64, 36)):
        b = pkg.synth.make_batch(16, T, N, dataset=ds, seed=5)
        v = pkg.synth.DATASETS[ds]
        assert (b["categories"][:, :, 0] == v["cls"]).all()
        assert b["lengths"][0] == T and (b["lengths"] >= 2).all() and (b["lengths"] <= T).all()
        for i in range(16):
            ln = int(b["lengths"][i])
            assert (b["frame_types"][i, :ln] != 0).all() and (b["frame_types"][i, ln:] == 0).all()
            assert b["frame_types"][i, ln - 1] == v["extract"]
            assert (b["categories"][i, ln - 1:, 1:] == 0).all()
        assert torch.equal(b["src_key_padding_mask_boxes"], b["categories"] == 0)
        assert ("scores" in b) == (ds == "action_genome")
        assert (b["boxes"][..., 0] <= b["boxes"][..., 2]).all() and (b["boxes"][..., 1] <= b["boxes"][..., 3]).all()


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/stlt_hip.h must serve a C host: compile it as C99 (pedantic), then build a C program against the shared
    library that calls two host-only entry points (no GPU needed) and check what it prints."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    importlib.import_module("__graft_entry__").build()
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, PKG_NAME)
    src = tmp_path / "client.c"
    src.write_text('#include <stdio.h>\n#include "stlt_hip.h"\n'
                   'int main(void) {\n'
                   '  stlt_params p; stlt_inputs in; stlt_opt_chunk c; (void)p; (void)in; (void)c;\n'
                   '  printf("%d %zu %zu %d\\n", stlt_version(), stlt_workspace_bytes(8, 32, 7, 768, 174), stlt_gemm_scratch_bytes(),\n'
                   '         stlt_linear_fwd(NULL, 0, NULL, NULL, NULL, 0, 1, 1, 32, 0, NULL));\n'
                   '  printf("%s\\n", stlt_last_error());\n  return 0;\n}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], check=True)
    exe = tmp_path / "client"
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-L", libdir, "-l:libstlt_hip.so", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    ver, ws, sk, rc = out[0].split()
    pkg = importlib.import_module(PKG_NAME)
    assert int(ver) == 110 and int(ws) == pkg.ops.workspace_bytes(8, 32, 7, 768, 174) and int(sk) == 64 * 1024 * 1024
    assert int(rc) == -1 and "null" in out[1]  # argument error reported through the C-ABI, message available


def test_trainer_fit_refuses_a_global_batch_that_does_not_divide_over_the_ranks():
    """Equal shards are what makes the per-rank mean losses / 1/world gradient average equal the global-batch step; an
    uneven (or empty) shard must be an error, not a silently different optimisation step or a hang in the all-reduce."""
    import importlib
    import pytest
    import torch
    T = importlib.import_module("revisiting-spatial-temporal-layouts_amd.train")
    tr = T.Trainer(torch.nn.Linear(4, 2), "something", world=4, rank=1, fused_optimizer=False)
    batch = {"categories": torch.zeros(6, 3, 2, dtype=torch.int64), "labels": torch.zeros(6, dtype=torch.int64)}
    with pytest.raises(ValueError, match="does not divide"):
        tr.fit([batch], "cpu")


def test_every_environment_knob_is_documented():
    """INTEGRATION.md section 2 lists every STLT_* environment variable the package and bench.py read (one table, with defaults)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = set()
    srcs = [os.path.join(root, "bench.py")]
    for dp, _, fs in os.walk(os.path.join(root, "revisiting-spatial-temporal-layouts_amd")):
        srcs += [os.path.join(dp, f) for f in fs if f.endswith((".hip", ".h", ".py"))]
    for path in srcs:
        text = open(path, errors="ignore").read()
        names |= set(re.findall(r'getenv\("(STLT_[A-Z0-9_]+)"\)', text))
        names |= set(re.findall(r'environ(?:\.get\(|\[)"(STLT_[A-Z0-9_]+)"', text))
    assert len(names) >= 25
    missing = sorted(n for n in names if n not in doc)
    assert not missing, f"undocumented environment knobs: {missing}"
