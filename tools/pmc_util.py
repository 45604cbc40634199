#!/usr/bin/env python3
"""Per-kernel-family utilisation figures from one rocprofv3 --pmc pass (SQ + GRBM counters) joined with the kernel
trace of the same run:

    python tools/pmc_util.py <out.json> <dir_with_pmc_pass>

effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (the chip lowers its clock under MFMA load: the 157.3
TFLOP/s f32 peak is quoted at 2.4 GHz); MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI cycles);
LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Guide: /opt/skills/guides/MI355X_MICROARCH.md (PMC slots,
DVFS give-back)."""
import csv, glob, json, os, re, sys


def family(name):
    m = re.search(r"attn_core_kernel<(?:true|false), (true|false), (true|false)>", name)
    if m:  # <STAMP, VARLEN, CAUSAL>
        return "attn_core_kernel/" + ("ragged-" if m.group(1) == "true" else "") + ("temporal" if m.group(2) == "true" else "spatial")
    m = re.search(r"attn16_kernel<(\d), (true|false), (true|false)", name)
    if m:  # <NB, FULL, CAUSAL, SPLIT>: 16-row tiles, the kernel of the L <= 64 passes (temporal = causal)
        return "attn16_kernel/" + ("temporal" if m.group(3) == "true" else "spatial")
    m = re.search(r"(gemm_nt_kernel<\d|gemm_fixup_kernel|[a-z_0-9]+_kernel)", name)
    return m.group(1) if m else name[:40]


def main():
    out, d = sys.argv[1:3]
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        raise SystemExit("no counter_collection.csv")
    acc = {}
    for f in cc:
        for r in csv.DictReader(open(f)):
            k = family(r["Kernel_Name"])
            a = acc.setdefault(k, {})
            a.setdefault(r["Counter_Name"], [0.0, 0])
            a[r["Counter_Name"]][0] += float(r["Counter_Value"])
            a[r["Counter_Name"]][1] += 1
            if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                a.setdefault("_ns", [0.0, 0])
                a["_ns"][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a["_ns"][1] += 1
    res = {}
    for k, a in acc.items():
        g = a.get("GRBM_GUI_ACTIVE")
        if not g or g[1] == 0:
            continue
        cyc = g[0] / 8.0  # sum over 8 XCDs -> chip cycles, summed over the launches
        e = {"launches": g[1], "gui_cycles_per_launch": round(cyc / g[1])}
        if "_ns" in a and a["_ns"][0] > 0:
            e["effective_clock_GHz"] = round(cyc / a["_ns"][0], 3)
            e["avg_us_per_launch_profiled"] = round(a["_ns"][0] / a["_ns"][1] / 1e3, 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            e["mfma_pipe_busy"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024.0 * cyc), 4)
        if "SQ_LDS_IDX_ACTIVE" in a and a["SQ_LDS_IDX_ACTIVE"][0] > 0:
            e["lds_bank_conflict_share"] = round(a.get("SQ_LDS_BANK_CONFLICT", [0.0])[0] / a["SQ_LDS_IDX_ACTIVE"][0], 5)
        w = a.get("SQ_WAVE_CYCLES")
        if w and w[0] > 0:
            for c, nm in (("SQ_WAIT_ANY", "wave_cycles_parked"), ("SQ_WAIT_INST_ANY", "wave_cycles_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing")):
                if c in a:
                    e[nm] = round(a[c][0] / w[0], 4)
        res[k] = e
    json.dump({"command": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                          "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
                          "--no-cpu-baseline --no-skip-padding  [cfg2, 1024 clips]", "kernels": res}, open(out, "w"), indent=1)
    for k in sorted(res):
        if "gemm_nt" in k or "attn_core" in k or "attn16" in k:
            print(k, res[k])


if __name__ == "__main__":
    main()
