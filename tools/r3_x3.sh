#!/bin/bash
mkdir -p gpurun_out/x3
timeout 900 python bench.py --no-side-legs > gpurun_out/x3/bench.json 2> gpurun_out/x3/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/x3/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["logit_max_abs_diff"]); print(j.get("split_bf16"))
PY
