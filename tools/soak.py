#!/usr/bin/env python3
"""Soak run of the kernels whose correctness rests on hand-written synchronisation (LDS-DMA with counted waits, loader / MFMA wave
barriers, stream-K partial tiles, the two-stream reverse sweep): each case is launched `--reps` times on bench-sized inputs, every result
must equal the first one bit for bit and stay within tolerance of a reference computed once (fp64 on the host for the small cases, the
other kernel path on the device for the large ones).  Found the stale bias strip of gemm16.hip (profiles/round4_gemm16_bias_strip_race.txt:
one failure in ~900 launches); kept as the tool to run after touching any pipeline.

    python tools/soak.py [--reps 30] [--only gemm16,mhsa,...]      -> one JSON line per case, exit code 1 on any mismatch
"""
import argparse, importlib, json, math, os, sys, time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("revisiting-spatial-temporal-layouts_amd")
ops, L = pkg.ops, pkg._lib
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def loop(name, fn, reps, ref=None, tol=0.0):
    first, bad_bits, bad_ref, worst = None, 0, 0, 0.0
    t0 = time.time()
    for r in range(reps):
        out = fn()
        outs = out if isinstance(out, (tuple, list)) else (out,)
        if first is None:
            first = [o.clone() for o in outs]
            if ref is not None:
                refs = ref if isinstance(ref, (tuple, list)) else (ref,)
                for o, want in zip(outs, refs):
                    worst = max(worst, (o - want).abs().max().item())
                if worst > tol:
                    bad_ref += 1
        else:
            if not all(torch.equal(o, f) for o, f in zip(outs, first)):
                bad_bits += 1
    torch.cuda.synchronize()
    rec = {"case": name, "reps": reps, "not_bit_identical": bad_bits, "out_of_tolerance": bad_ref, "max_err": worst, "tol": tol, "s": round(time.time() - t0, 2)}
    print(json.dumps(rec), flush=True)
    return bad_bits + bad_ref


def gemm16_cases(reps):
    bad = 0
    for M, N, K in ((33000, 768, 128), (33000, 768, 64), (20000, 2304, 768), (4096, 3072, 768), (2048, 768, 3072), (33000, 144, 96)):
        x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K))
        b = (torch.arange(N, dtype=torch.float32, device=DEV) - N / 2) * 0.25
        r = rnd(M, N, seed=3)
        with ops.gemm_scratch(DEV):
            ops.set_gemm_small_tiles(0)
            ref = ops.linear(x, w, b)       # the large-tile kernel
            ops.set_gemm_small_tiles(-2)
        tol = 3e-5 * math.sqrt(K) * max(1.0, ref.abs().max().item()) / 10
        for tr, tc in ops.SMALL_TILES:
            bad += loop(f"gemm16 fwd {M}x{N}x{K} t{tr}x{tc} bias", lambda: ops.linear_small(x, w, b, tc, tile_rows=tr), reps, ref, tol)
            bad += loop(f"gemm16 fwd {M}x{N}x{K} t{tr}x{tc} bias+gelu", lambda: ops.linear_small(x, w, b, tc, act=1, tile_rows=tr), max(reps // 3, 3))
            bad += loop(f"gemm16 fwd {M}x{N}x{K} t{tr}x{tc} bias+residual", lambda: ops.linear_small(x, w, b, tc, residual=r, tile_rows=tr), max(reps // 3, 3), ref + r, tol)
    for M, n_out, k_in in ((33000, 128, 768), (20000, 768, 768), (4096, 768, 3072), (2048, 2304, 768)):
        dy, w, r = rnd(M, n_out, seed=4), rnd(n_out, k_in, seed=5, scale=1 / math.sqrt(n_out)), rnd(M, k_in, seed=6)
        with ops.gemm_scratch(DEV):
            ref = ops.gemm(dy, w, trans_b=True)
        tol = 3e-5 * math.sqrt(n_out) * max(1.0, ref.abs().max().item()) / 10
        for tr, tc in ((128, 48), (128, 96), (128, 192), (64, 64), (64, 160), (64, 256), (32, 128), (32, 256)):
            bad += loop(f"gemm16 dX {M}x{n_out}->{k_in} t{tr}x{tc}", lambda: ops.input_grad_small(dy, w, tc, tile_rows=tr), reps, ref, tol)
            bad += loop(f"gemm16 dX {M}x{n_out}->{k_in} t{tr}x{tc} +residual", lambda: ops.input_grad_small(dy, w, tc, residual=r, tile_rows=tr), max(reps // 3, 3), ref + r, tol)
    return bad


def gemm_cases(reps):
    bad = 0
    for M, N, K in ((14336, 768, 768), (14336, 3072, 768), (2048, 768, 3072), (229376, 768, 768), (4000, 174, 768)):
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3)
        ops.set_gemm_small_tiles(0)
        try:
            plain = ops.linear(x, w, b)  # no scratch lent: whole tiles
            with ops.gemm_scratch(DEV):
                tol = 3e-5 * math.sqrt(K) * max(1.0, plain.abs().max().item()) / 10
                bad += loop(f"gemm stream-K fwd {M}x{N}x{K}", lambda: ops.linear(x, w, b), reps, plain, tol)
                bad += loop(f"gemm stream-K fwd+gelu {M}x{N}x{K}", lambda: ops.linear(x, w, b, act=1), max(reps // 3, 3))
                dy = rnd(M, N, seed=7)
                bad += loop(f"gemm stream-K dX {M}x{N}->{K}", lambda: ops.gemm(dy, w, trans_b=True), reps)
                if M % 32 == 0:
                    bad += loop(f"gemm stream-K dW {N}x{K} over {M} rows", lambda: ops.gemm(dy, x, trans_a=True, trans_b=True), reps)
        finally:
            ops.set_gemm_small_tiles(-2)
    return bad


def gemm_tiny_cases(reps):
    """gemm.hip on the shapes of a 4-clip forward (896 / 128 / 4 rows): a handful of tiles, stream-K over a few dozen k-steps with
    workgroups of ~4 k-steps each, segments of a single k-step, the fix-up launch."""
    bad = 0
    ops.set_gemm_small_tiles(0)
    try:
        for M in (4, 100, 128, 896, 1024):
            for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (174, 768)):
                x, w, b, r = rnd(M, K, seed=M + N), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3), rnd(M, N, seed=4)
                plain = ops.linear(x, w, b)
                tol = 3e-6 * math.sqrt(K) * max(1.0, plain.abs().max().item())
                with ops.gemm_scratch(DEV):
                    bad += loop(f"gemm tiny stream-K fwd {M}x{N}x{K}", lambda: ops.linear(x, w, b), reps, plain, tol)
                    bad += loop(f"gemm tiny stream-K fwd+gelu {M}x{N}x{K}", lambda: ops.linear(x, w, b, act=1), max(reps // 2, 3))
                    dy = rnd(M, N, seed=7)
                    bad += loop(f"gemm tiny stream-K dX {M}x{N}->{K}", lambda: ops.gemm(dy, w, trans_b=True), max(reps // 2, 3))
                    bad += loop(f"gemm tiny stream-K dX+R {M}x{N}->{K}", lambda: ops.gemm(dy, w, trans_b=True, add=x), max(reps // 2, 3))
    finally:
        ops.set_gemm_small_tiles(-2)
    return bad


def small_forward_cases(reps):
    """The golden-sized forwards (2 - 8 clips): what the parity tests launch."""
    bad = 0
    for name, B in (("cfg2", 4), ("cfg1", 8), ("cfg2p", 3), ("cfg4", 2), ("refdef", 4)):
        c = pkg.synth.CONFIGS[name]
        m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
        m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234))
        m.to(DEV).train(False)
        batch = {k: v.to(DEV) for k, v in pkg.synth.make_batch(B, c["T"], c["N"], seed=0).items()}
        for cls_only in (True, False):
            m.backbone.cls_only_last_spatial = m.backbone.last_row_only_temporal = cls_only
            with torch.no_grad():
                bad += loop(f"forward {name} B={B} cls_only={cls_only}", lambda: m(batch)["stlt"], reps)
    return bad


def mhsa_cases(reps):
    bad = 0
    H, d = 12, 768
    w, b = rnd(3 * d, d, seed=1, scale=2 / math.sqrt(d)), rnd(3 * d, seed=2, scale=0.5)
    for S, Lq, causal in ((1024, 32, True), (1024, 17, True), (256, 64, True), (1030, 24, True), (8192, 7, False), (8192, 5, False), (4096, 8, False), (700, 36, False)):
        x = rnd(S, Lq, d, seed=S + Lq, scale=1.5)
        kpm = (torch.rand(S, Lq, generator=torch.Generator().manual_seed(S)) < 0.3).to(DEV)
        kpm[:, 0] = False
        with ops.gemm_scratch(DEV):
            qkv = ops.linear(x.view(S * Lq, d), w, b).view(S, Lq, 3 * d)
            two = ops.attn_core(qkv, kpm, causal, H)
        bad += loop(f"mhsa fused S={S} L={Lq} causal={causal}", lambda: ops.mhsa_fused(x, w, b, kpm, H, causal=causal), reps, two, 2e-5 * max(1.0, two.abs().max().item()))
        bad += loop(f"mhsa fused (training form, p=0.1) S={S} L={Lq} causal={causal}",
                    lambda: ops.mhsa_fused(x, w, b, kpm, H, causal=causal, want_qkv=True, dropout_p=0.1, seed=11, site=3), max(reps // 2, 3))
        bad += loop(f"attn16 / attn core S={S} L={Lq} causal={causal}", lambda: ops.attn_core(qkv, kpm, causal, H), reps)
        g = rnd(S, Lq, d, seed=9)
        bad += loop(f"attn backward S={S} L={Lq} causal={causal} p=0.1", lambda: ops.attn_core_bwd(qkv, g, kpm, causal, H, 0.1, 5, 7, want_bias_grad=True), max(reps // 2, 3))
    return bad


def block_cases(reps):
    """The fusion models' block-level calls at their row counts: forward + backward of a feed-forward block (the GELU backward rides in the
    small-tile input-gradient product's epilogue) and of self- / cross-attention blocks, same seed every repetition."""
    bad = 0
    d, H = 768, 12
    for S, Lq in ((64, 32), (64, 33)):
        M = S * Lq
        x, g = rnd(S, Lq, d, seed=1), rnd(S, Lq, d, seed=2)
        c = rnd(S, 33, d, seed=3)
        w1, b1 = rnd(4 * d, d, seed=3, scale=1 / math.sqrt(d)), rnd(4 * d, seed=4, scale=0.1)
        w2, b2 = rnd(d, 4 * d, seed=5, scale=1 / math.sqrt(4 * d)), rnd(d, seed=6, scale=0.1)
        w_in, b_in = rnd(3 * d, d, seed=7, scale=1 / math.sqrt(d)), rnd(3 * d, seed=8, scale=0.1)
        w_o, b_o = rnd(d, d, seed=9, scale=1 / math.sqrt(d)), rnd(d, seed=10, scale=0.1)
        ln_w, ln_b = 1 + 0.1 * rnd(d, seed=11), 0.1 * rnd(d, seed=12)

        def ffn():
            torch.manual_seed(5)
            leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, ln_w, ln_b)]
            out = ops.FfnBlockFn.apply(leaves[0], 1e-5, L.ACT_GELU, True, 0.1, *leaves[1:])
            out.backward(g)
            return [out.detach()] + [t.grad for t in leaves]

        def attn(cross):
            def f():
                torch.manual_seed(6)
                leaves = [t.clone().requires_grad_(True) for t in (x, c, w_in, b_in, w_o, b_o, ln_w, ln_b)]
                out = ops.AttnBlockFn.apply(leaves[0], leaves[1] if cross else None, None, not cross, H, 1e-5, 0.1, *leaves[2:])
                out.backward(g)
                return [out.detach()] + [t.grad for i, t in enumerate(leaves) if cross or i != 1]
            return f

        bad += loop(f"ffn block fwd+bwd {M} rows p=0.1", ffn, max(reps // 4, 4))
        bad += loop(f"self-attention block fwd+bwd {M} rows p=0.1", attn(False), max(reps // 4, 4))
        bad += loop(f"cross-attention block fwd+bwd {M} x {S * 33} rows p=0.1", attn(True), max(reps // 4, 4))
    return bad


def train_cases(reps):
    bad = 0
    for name, B in (("cfg2", 64), ("cfg2p", 16), ("cfg1", 256)):
        for side in (True, False):
            c = pkg.synth.CONFIGS[name]
            torch.manual_seed(0)
            m = pkg.Stlt(pkg.StltModelConfig(**dict(pkg.synth.model_kwargs(name), hidden_dropout_prob=0.1)))
            sd = pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3)
            m.load_state_dict(sd)
            m.to(DEV).train(True)
            batch = {k: v.to(DEV) for k, v in pkg.synth.make_batch(B, c["T"], c["N"], seed=4, min_len=2).items()}
            labels = torch.randint(0, c["num_classes"], (B,), generator=torch.Generator().manual_seed(1)).to(DEV)
            ops.set_train_side_stream(side)

            def step():
                for p in m.parameters():
                    p.grad = None
                torch.manual_seed(7)  # same dropout masks every repetition
                loss = torch.nn.functional.cross_entropy(m(batch)["stlt"], labels)
                loss.backward()
                return [loss.detach()] + [p.grad for p in m.parameters() if p.grad is not None]

            bad += loop(f"train forward+backward {name} B={B} side_stream={side}", step, max(reps // 3, 4))
            ops.set_train_side_stream(None)  # back to STLT_TRAIN_DW_STREAM / the default
    return bad


def forward_cases(reps):
    bad = 0
    for name, B in (("cfg2", 1024), ("cfg2p", 256), ("refdef", 1024), ("cfg4", 64)):
        c = pkg.synth.CONFIGS[name]
        m = pkg.Stlt(pkg.StltModelConfig(**pkg.synth.model_kwargs(name)))
        m.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3))
        m.to(DEV).train(False)
        batch = {k: v.to(DEV) for k, v in pkg.synth.make_batch(B, c["T"], c["N"], seed=4, min_len=2).items()}
        for skip in (False, True):
            m.backbone.skip_padding = skip
            with torch.no_grad():
                bad += loop(f"forward {name} B={B} skip_padding={skip}", lambda: m(batch)["stlt"], max(reps // 3, 4))
    return bad


CASES = {"gemm16": gemm16_cases, "gemm": gemm_cases, "gemm_tiny": gemm_tiny_cases, "small_forward": small_forward_cases, "mhsa": mhsa_cases, "blocks": block_cases, "train": train_cases, "forward": forward_cases}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", default=",".join(CASES))
    a = ap.parse_args()
    total = 0
    for k in a.only.split(","):
        total += CASES[k](a.reps)
    print(json.dumps({"soak": "done", "mismatching_cases": total}))
    sys.exit(1 if total else 0)
