#!/usr/bin/env python3
"""Aggregate rocprofv3 counter-collection CSVs (one pass per counter) into profiles/<tag>_traffic_pmc.json.

    python tools/pmc_traffic.py <out.json> <dir_with_FETCH_SIZE_pass> <dir_with_WRITE_SIZE_pass>

Per kernel family: average FETCH_SIZE / WRITE_SIZE per launch (KB as rocprofv3 reports them) and the HBM-side bytes
per launch = FETCH_SIZE*2*1024 + WRITE_SIZE*1024 (gfx950 tallies 128-B fabric read requests as 64 B:
/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section).  Warm-up launches are included; they move the
averages by < 1 %."""
import csv, glob, json, os, re, sys


def family(name: str) -> str:
    m = re.search(r"attn_core_kernel<(?:true|false), (true|false), (true|false)>", name)
    if m:  # <STAMP, VARLEN, CAUSAL>: the temporal (causal) and spatial passes are distinct symbols
        return "attn_core_kernel/" + ("ragged-" if m.group(1) == "true" else "") + ("temporal" if m.group(2) == "true" else "spatial")
    m = re.search(r"attn16_kernel<(\d), (true|false), (true|false)", name)
    if m:  # <NB, FULL, CAUSAL, SPLIT>: 16-row tiles, the kernel of the L <= 64 passes (temporal = causal)
        return "attn16_kernel/" + ("temporal" if m.group(3) == "true" else "spatial")
    m = re.search(r"mhsa16_kernel<(\d), (true|false), (true|false)(?:, (?:true|false))?>", name)
    if m:  # <NKB, CAUSAL, TRAIN, WINDOW>: the fused in-projection + attention kernel (temporal = causal)
        return "mhsa16_kernel/" + ("temporal" if m.group(2) == "true" else "spatial")
    m = re.search(r"(gemm_nt_kernel<\d|gemm_fixup_kernel|[a-z_0-9]+_kernel)", name)
    return m.group(1) if m else name[:40]


def read_pass(d: str, counter: str):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = family(r["Kernel_Name"])
            tot, n = acc.get(k, (0.0, 0))
            acc[k] = (tot + float(r["Counter_Value"]), n + 1)
    return acc


def main():
    out, d_fetch, d_write = sys.argv[1:4]
    fe, wr = read_pass(d_fetch, "FETCH_SIZE"), read_pass(d_write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f_tot, f_n = fe.get(k, (0.0, 0))
        w_tot, w_n = wr.get(k, (0.0, 0))
        f_kb = f_tot / f_n if f_n else 0.0
        w_kb = w_tot / w_n if w_n else 0.0
        kernels[k] = {"FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
                      "launches_profiled": max(f_n, w_n), "hbm_bytes_per_launch_corrected": int(f_kb * 2 * 1024 + w_kb * 1024)}

    def avg(prefix):
        tot = n = 0
        for k, v in kernels.items():
            if k.startswith(prefix):
                tot += v["hbm_bytes_per_launch_corrected"] * v["launches_profiled"]
                n += v["launches_profiled"]
        return int(tot / n) if n else None

    doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) --output-format csv -- "
                      "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-padding  [cfg2, per-GPU batch 1024]",
           "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B fabric requests at 64 B, MI355X_MICROARCH.md HBM section); "
                         "WRITE_SIZE as read; 1 KB = 1024 B; L2 memory-side requests, Infinity-Cache hits included",
           "gemm_avg_bytes_per_launch": avg("gemm_nt_kernel"),
           "attn_temporal_avg_bytes_per_launch": (avg("attn16_kernel/temporal") or avg("attn_core_kernel/temporal")), "attn_temporal_algorithmic_bytes_per_launch": 1024 * (16 * 32 * 768 + 32),
           "attn_spatial_avg_bytes_per_launch": (avg("attn16_kernel/spatial") or avg("attn_core_kernel/spatial")), "attn_spatial_algorithmic_bytes_per_launch": 1024 * (16 * 32 * 7 * 768 + 32 * 7),
           "mhsa_fused_avg_bytes_per_launch": avg("mhsa16_kernel/temporal"), "mhsa_fused_algorithmic_bytes_per_launch": 1024 * 32 * 768 * 8 + 4 * (3 * 768 * 768 + 3 * 768) + 1024 * 32,
           "mhsa_fused_spatial_avg_bytes_per_launch": avg("mhsa16_kernel/spatial"),
           "mhsa_fused_spatial_algorithmic_bytes_per_launch": 1024 * 32 * 7 * 768 * 8 + 4 * (3 * 768 * 768 + 3 * 768) + 1024 * 32 * 7,
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("gemm_avg_bytes_per_launch", "attn_temporal_avg_bytes_per_launch", "attn_spatial_avg_bytes_per_launch")}))


if __name__ == "__main__":
    main()
