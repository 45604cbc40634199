#!/bin/bash
# round 3, row N1: parity of the fused MHSA kernel, stand-alone A/B per clip count, whole-forward A/B at 1024 clips, and the
# FETCH_SIZE / WRITE_SIZE counters of both paths (separate --pmc passes).  Writes gpurun_out/r3_mhsa/.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_mhsa; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mhsa" > $O/pytest_mhsa.log 2>&1; tail -3 $O/pytest_mhsa.log
python tools/bench_mhsa.py > $O/bench_mhsa.jsonl 2> $O/bench_mhsa.err; cat $O/bench_mhsa.jsonl
for k in 0 1; do
  STLT_FUSED_MHSA=$k python bench.py --no-cpu-baseline --no-skip-padding --no-side-legs > $O/fwd1024_fused$k.json 2> $O/fwd1024_fused$k.err
  python - $O/fwd1024_fused$k.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], j["value"], "clips/s", j["ms_per_step"], "ms", {k: j["kernel_ms_per_step"][k] for k in ("gemm", "attn_temporal")}, j["roofline"]["frac"])
PY
done
STLT_FUSED_MHSA=1 timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or logits or cfg2" > $O/pytest_model_fused.log 2>&1; tail -2 $O/pytest_model_fused.log
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for which in two fused; do
    rm -rf /tmp/pmc_$c_$which
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${c}_$which -o o -- python3 $R/tools/bench_mhsa.py --clips 1024 --iters 5 --only $which > /dev/null 2>&1
  done
done
python3 - <<'PY' > $O/pmc_traffic.json
import csv, glob, json, re
out = {}
for which in ("two", "fused"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = {}
        for f in glob.glob(f"/tmp/pmc_{c}_{which}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != c: continue
                n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60]
                t, k = acc.get(n, (0.0, 0)); acc[n] = (t + float(r["Counter_Value"]), k + 1)
        for n, (t, k) in acc.items():
            if "gemm_nt" in n or "attn16" in n or "mhsa" in n or "attn_core" in n:
                out.setdefault(which, {}).setdefault(n, {})[c + "_KB_per_launch"] = round(t / k, 1)
for which in out:
    for n, v in out[which].items():
        v["memory_side_MB_per_launch_corrected"] = round((v.get("FETCH_SIZE_KB_per_launch", 0) * 2 + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024 / 1e6, 1)
print(json.dumps(out, indent=1))
PY
cat $O/pmc_traffic.json
