#!/usr/bin/env python3
"""Every launch of a step against its own bound (round-5 review, item 1: "neither of us can say whether 17.9 k is 0.80 or 0.95 of what the
shapes allow").

    python tools/launch_bound.py --config refdef --batch 64                  # the reference's default layout at its own batch size
    python tools/launch_bound.py --config cfg2 --batch 64 --mode train       # BASELINE config 3's per-GPU shard

The library's recorder (stlt_prof_enable / stlt_prof_launches) brackets every launch with two events on the launch stream and keeps the
launcher's own description of it: shape, tile, workgroups, rounds, k-steps, declared FLOPs / algorithmic bytes.  Per launch position of
the step (averaged over --steps replays):

  measured us   event time of the launch (a stream-K product includes its fix-up launch); two events per launch cost about 2 us, so the
                events' sum overstates the step.  With --trace (two passes, see below) the column is rocprofv3's: the launch's kernel time
                plus the idle gap up to the next launch's first kernel — the launch's share of the wall clock, the sum IS the step.
  bound us      matrix-core launches:  min over the tile shapes the library has (256x128 and the 15 small tiles) of
                    rounds x k-steps x t_kstep(tile) + epilogue + BOUNDARY
                with rounds = ceil(tiles / 256 CUs), t_kstep = 2 tm tn 32 / (0.94 x 157.3 TF / 256) — the rate the 128 x 192 tile's
                k-loop reaches (profiles/round5_gemm16_ablation.txt) granted to EVERY tile —, epilogue = the tile's output (+ add-source)
                bytes at a CU's share of 6.3 TB/s, BOUNDARY = 1.5 us (launch + drain between two dependent kernels of one stream);
                grouped weight-gradient launches: total k-steps of 256x128 tiles spread evenly over the CUs;
                fused MHSA: rounds x (the item's product k-steps + its attention FLOPs at the same rate) + BOUNDARY;
                HBM-bound kernels:  declared algorithmic bytes / 6.3 TB/s + BOUNDARY;  launches that declare nothing: BOUNDARY.
  bound / measured   1.00 = at the bound.  The step's figure is sum(bound) / sum(measured); "wall" is the clock around the whole step.
Two passes for rocprofv3 durations (the recorder's events would perturb the gaps, so the traced replays run with the recorder off):
    rocprofv3 --kernel-trace --output-format csv -d /tmp/lb -o lb -- python3 tools/launch_bound.py --config refdef --batch 64 --notes /tmp/lb/notes.json
    python tools/launch_bound.py --merge /tmp/lb/notes.json --trace $(find /tmp/lb -name '*kernel_trace.csv')
The bound deliberately ignores what makes narrow tiles slower per k-step (16 flop per staged byte against 48 for 128 x 192): it is what
the shapes allow on this chip, not what this kernel design allows.
"""
import argparse
import importlib
import math
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "revisiting-spatial-temporal-layouts_amd"

CUS = 256
PEAK_CU = 157.3e12 / CUS          # f32-MFMA dense peak per CU (MI355X_MICROARCH.md)
PIPE = 0.94                       # of the pipe: the best tile's k-loop
HBM = 6.3e12                      # achievable HBM3E stream rate (bytes/s), the guide's measured figure
BOUNDARY = 1.5                    # us
TILES = [(256, 128)] + [(128, c) for c in (48, 64, 96, 128, 144, 192)] + [(64, c) for c in (64, 96, 128, 160, 192, 256)] + [(32, c) for c in (128, 192, 256)]


def t_kstep_us(tm, tn):
    return 2.0 * tm * tn * 32 / (PIPE * PEAK_CU) * 1e6


def product_bound(M, N, K, add=False):
    best = None
    for tm, tn in TILES:
        tiles = math.ceil(M / tm) * math.ceil(N / tn)
        rounds = math.ceil(tiles / CUS)
        epi = tm * tn * 4 * (2 if add else 1) / (HBM / CUS) * 1e6
        us = rounds * (K / 32) * t_kstep_us(tm, tn) + epi + BOUNDARY
        if best is None or us < best[0]:
            best = (us, f"{tm}x{tn} r{rounds}")
    return best


def bound_of(rec):
    note = rec["note"]
    g = lambda key, cast=int: cast(re.search(rf"\b{key}=(-?\d+)", note).group(1))  # noqa: E731
    if note.startswith("gemm(dW group)"):
        ksteps = g("ksteps")
        return ksteps / CUS * t_kstep_us(256, 128) + 256 * 128 * 8 / (HBM / CUS) * 1e6 + BOUNDARY, "dW group"
    if note.startswith("gemm"):
        us, how = product_bound(g("M"), g("N"), g("K"), "+R" in note)
        return us, how
    if note.startswith("mhsa16"):
        rounds, L, d, rows = g("rounds"), g("L"), g("d"), int(re.search(r"item=(\d+)x192", note).group(1))
        item_flops = 2.0 * rows * 192 * d + 4.0 * (rows // L) * L * L * 64
        return rounds * item_flops / (PIPE * PEAK_CU) * 1e6 + BOUNDARY, f"items r{rounds}"
    if rec["bytes"] > 0:
        return rec["bytes"] / HBM * 1e6 + BOUNDARY, "hbm"
    return BOUNDARY, "-"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="refdef")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--mode", choices=("forward", "train", "cacnf_train"), default="forward")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--skip-padding", action="store_true")
    ap.add_argument("--notes", default=None, help="pass 1 (under rocprofv3): write the launches' notes here and end with --steps replays that the trace will hold last")
    ap.add_argument("--merge", default=None, help="pass 2: the notes file of pass 1 ...")
    ap.add_argument("--trace", default=None, help="... and rocprofv3's kernel trace (csv) of pass 1")
    args = ap.parse_args()
    if args.merge:
        return merge(args)
    import torch
    pkg = importlib.import_module(PKG)
    dev = torch.device("cuda", 0)
    c = pkg.synth.CONFIGS[args.config]
    B, T, N = args.batch, c["T"], c["N"]
    train = args.mode != "forward"
    kw = pkg.synth.model_kwargs(args.config)
    if train:
        kw["hidden_dropout_prob"] = 0.1
    if args.mode == "cacnf_train":
        model = pkg.models_factory["cacnf"](pkg.MultimodalModelConfig(**dict(kw, appearance_num_frames=32)))
    else:
        model = pkg.Stlt(pkg.StltModelConfig(**kw))
    model.load_state_dict(pkg.synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234))
    model.train(train).to(dev)
    batch = pkg.synth.make_batch(B, T, N, dataset=c["dataset"], seed=2000)
    if args.mode == "cacnf_train":
        batch["appearance_features"] = pkg.synth.make_appearance_features(B, seed=1)
    batch["labels"] = torch.randint(0, c["num_classes"], (B,), generator=torch.Generator().manual_seed(0))
    batch = {k: v.to(dev) for k, v in batch.items()}
    if train:
        tr = pkg.train.Trainer(model, "something", learning_rate=5e-5, weight_decay=1e-3, clip_val=5.0, warmup_steps=2, total_steps=100000)
        step = lambda: tr.step(batch)  # noqa: E731
    else:
        model.backbone.skip_padding = args.skip_padding

        def step():
            with torch.no_grad():
                return model(batch)["stlt"]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    wall_us = (time.perf_counter() - t0) / args.steps * 1e6
    side_was = pkg.ops.get_train_side_stream()
    pkg.ops.set_train_side_stream(False)  # per-launch events: overlapping spans would not add up to the step
    pkg.ops.prof_enable(True)
    try:
        pkg.ops.prof_launches()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        recs = pkg.ops.prof_launches()
    finally:
        pkg.ops.prof_enable(False)
        pkg.ops.set_train_side_stream(side_was)
    per = len(recs) // args.steps
    assert per * args.steps == len(recs), (len(recs), args.steps)
    rows = []
    for i in range(per):
        same = recs[i::per]
        assert all(r["note"] == same[0]["note"] and r["kernel"] == same[0]["kernel"] and r["kernels"] == same[0]["kernels"] for r in same), (i, same[0], same[1])
        us = sum(r["us"] for r in same) / len(same)
        b, how = bound_of(same[0])
        rows.append((i, same[0]["kernel"], us, b, how, same[0]["note"], same[0]["flops"], same[0]["kernels"]))
    head = (f"# {args.mode} of {args.config} (T={T}, N={N}, d={c['hidden_size']}), {B} clips, {args.steps} replays; wall {wall_us:.1f} us per step"
            f"{' (side stream on; the per-launch replays run on one stream)' if train else ''}; {per} launches per step")
    if args.notes:
        import json
        # the trace's LAST kernels: a marker, then --steps replays with the recorder off (train: still on one stream, so that kernels do not overlap
        # and the trace's start order is the launch order: the weight copies are written on the caller's stream too)
        pkg.ops.set_train_side_stream(False)
        if train and tr.transposed is not None:
            tr.transposed._side = torch.cuda.current_stream(dev)
        torch.cuda.synchronize(dev)
        torch.arange(64, device=dev).flip(0)  # the marker: a kernel name nothing else in the process has
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        traced_wall = (time.perf_counter() - t0) / args.steps * 1e6
        pkg.ops.set_train_side_stream(side_was)
        with open(args.notes, "w") as f:
            json.dump({"head": head, "steps": args.steps, "wall_us": wall_us, "traced_wall_us": traced_wall, "rows": rows}, f)
        print(f"[launch_bound] notes of {per} launches per step -> {args.notes}; traced replays {traced_wall:.1f} us per step (under the profiler)")
        return
    print(head)
    report(rows, wall_us, "event")


def merge(args):
    """Pass 2: rocprofv3's kernel trace of pass 1 against its notes: per launch, kernel time + the gap to the next launch."""
    import csv
    import json
    with open(args.merge) as f:
        j = json.load(f)
    rows, steps = j["rows"], j["steps"]
    with open(args.trace) as f:
        disp = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(f)), key=lambda t: t[0])
    mark = max(i for i, d in enumerate(disp) if "flip" in d[2].lower())
    everything = disp[mark + 1:]
    # kernels torch itself launches inside a step (the trainer's buffer clears, loss sums, ...) are not the library's launches: their time is
    # part of the gap behind the launch they follow, and is reported on a line of its own
    is_foreign = lambda name: "at::" in name or name.startswith("__amd_rocclr_")  # noqa: E731  (torch's kernels; the runtime's memset / copy kernels)
    foreign = [d for d in everything if is_foreign(d[2])]
    tail = [d for d in everything if not is_foreign(d[2])]
    kps = sum(r[7] for r in rows)
    if len(tail) != kps * steps:
        from collections import Counter
        names = Counter(re.sub(r"<.*", "", d[2]) for d in tail)
        raise SystemExit(f"{len(tail)} library kernels after the marker, the notes say {kps} x {steps}: a launcher starts kernels the recorder does not count\n"
                         + "\n".join(f"  {n / steps:8.2f} per step  {k}" for k, n in names.most_common()) + "\nnotes: "
                         + ", ".join(f"{k} x{sum(r[7] for r in rows if r[1] == k)}" for k in sorted({r[1] for r in rows})))
    acc = [[0.0, 0.0] for _ in rows]
    for sidx in range(steps):
        ks = tail[sidx * kps:(sidx + 1) * kps]
        nxt = tail[(sidx + 1) * kps][0] if sidx + 1 < steps else None
        k = 0
        for li, r in enumerate(rows):
            mine = ks[k:k + r[7]]
            k += r[7]
            if not mine:
                continue
            busy = sum(e - s for s, e, _ in mine)
            after = ks[k][0] if k < len(ks) else nxt
            span_end = after if after is not None else mine[-1][1]
            acc[li][0] += busy / 1e3
            acc[li][1] += (span_end - mine[0][0]) / 1e3
    n_gap = steps  # the last step has no successor: its last launch is charged its kernel time only
    out = [(r[0], r[1], acc[i][1] / steps, r[3], r[4], r[5] + f"  [kernel {acc[i][0] / steps:.2f} us]", r[6], r[7]) for i, r in enumerate(rows)]
    step_span = (tail[-1][1] - tail[0][0]) / 1e3 / steps
    print(j["head"])
    print(f"# measured = rocprofv3 --kernel-trace of {steps} replays with the recorder off: kernel time + gap to the next launch; the replays' span in the trace "
          f"{step_span:.1f} us per step (wall clock outside the profiler {j['wall_us']:.1f} us)")
    if foreign:
        print(f"# kernels launched by torch / the runtime's memsets inside the steps (counted in the gaps above): {len(foreign) / steps:.1f} per step, {sum(e - s0 for s0, e, _ in foreign) / 1e3 / steps:.1f} us per step")
    report(out, step_span, "rocprofv3")


def report(rows, wall_us, how_measured):
    import math  # noqa: F401
    per = len(rows)
    if True:
        print(f"# bound: t_kstep at {PIPE} of {PEAK_CU / 1e9:.1f} GF/CU for every tile; HBM {HBM / 1e12:.1f} TB/s; boundary {BOUNDARY} us; measured by {how_measured}")
        print(f"{'#':>3} {'kernel':<18} {'meas us':>8} {'bound us':>8} {'b/m':>5}  {'best':<12} note")
        for i, k, us, b, how, note, *_ in rows:
            if us <= 0:  # a kernel outside the recorder's scopes in the event-timed form: counted, not timed
                print(f"{i:>3} {k:<18} {'-':>8} {b:>8.2f} {'-':>5}   {how:<12} {note}")
                continue
            flag = " <" if b / us < 0.85 else ""
            print(f"{i:>3} {k:<18} {us:>8.2f} {b:>8.2f} {b / us:>5.2f}{flag:<2} {how:<12} {note}")
        rows = [r for r in rows if r[2] > 0]
        tm, tb = sum(r[2] for r in rows), sum(r[3] for r in rows)
        print(f"# step: sum(measured) {tm:.1f} us, sum(bound) {tb:.1f} us, achieved / bound = {tb / tm:.3f}; step clock {wall_us:.1f} us = {wall_us / tm:.3f} x sum(measured); "
              f"bound / step clock = {tb / wall_us:.3f}")
        by = {}
        for _, k, us, b, *_ in rows:
            a = by.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1; a[1] += us; a[2] += b
        for k, (n, us, b) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            print(f"#   {k:<18} {n:>3} launches {us:>9.1f} us measured {b:>9.1f} us bound  {b / us:.3f}   gap {us - b:>8.1f} us ({(us - b) / tm * 100:.1f} % of the step)")
        low = [r for r in rows if r[3] / r[2] < 0.85]
        print(f"# launches under 0.85 of their bound: {len(low)} of {per}, {sum(r[2] - r[3] for r in low):.1f} us above their bounds in total")


if __name__ == "__main__":
    main()
