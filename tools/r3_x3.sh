#!/bin/bash
mkdir -p gpurun_out/x3
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/x3/bench.json 2> gpurun_out/x3/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/x3/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"]); print({k:v for k,v in j["split_bf16"].items() if k!="note"})
for k in ("train_step","cfg4","small_batch","cfg5"): print(k, j[k]["value"], j[k]["ms_per_step"], {a:b for a,b in j[k]["split_bf16"].items() if a!="note"})
PY
