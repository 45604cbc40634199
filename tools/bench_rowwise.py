#!/usr/bin/env python3
"""Stand-alone launches of the row-wise kernels (embedding, residual + LayerNorm) at the headline's and the 64-clip forward's row counts:
us per call (torch events around 50 calls, output buffer reused by the caching allocator) and the HBM rate of the algorithmic bytes.

    python tools/bench_rowwise.py                       # the library in the tree
    STLT_HIP_LIB=build/variants/libstlt_hip_<tag>.so STLT_EMBED_ROWS=0 python tools/bench_rowwise.py   # A/B builds / knobs
"""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


def main():
    c = pkg.synth.CONFIGS["cfg2"]
    d = c["hidden_size"]
    m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs("cfg2"))).to("cuda")
    emb = m.backbone.frames_embeddings.layout_embedding.category_box_embeddings
    out = {"lib": os.environ.get("STLT_HIP_LIB", "tree"), "STLT_EMBED_ROWS": os.environ.get("STLT_EMBED_ROWS", "default"), "embed": [], "add_ln": []}
    for B in (64, 147, 256, 1024):
        batch = {k: v.to("cuda") for k, v in pkg.synth.make_batch(B, c["T"], c["N"], dataset=c["dataset"], seed=1, with_scores=True).items()}
        args = (batch["categories"], batch["boxes"], batch["scores"], emb.category_embeddings.weight, emb.box_embedding.weight, emb.box_embedding.bias,
                emb.score_embeddings.weight, emb.score_embeddings.bias, emb.layer_norm.weight, emb.layer_norm.bias, 1e-12)
        us = timed(lambda: pkg.ops.embed(*args))
        n = B * c["T"] * c["N"]
        out["embed"].append({"clips": B, "rows": n, "us": round(us, 2), "TB_per_s": round(n * (4 * d + 29) / us / 1e6, 3)})
    ln_w, ln_b = emb.layer_norm.weight, emb.layer_norm.bias
    for rows in (64, 1024, 2048, 14336, 32768, 229376):
        x = torch.randn(rows, d, device="cuda")
        us = timed(lambda: pkg.ops.add_layernorm(x, None, ln_w, ln_b, 1e-12))
        out["add_ln"].append({"rows": rows, "us": round(us, 2), "TB_per_s": round(rows * 8 * d / us / 1e6, 3)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
