#!/bin/bash
mkdir -p gpurun_out/x3
timeout 1500 python -m pytest tests/test_gemm_bf16x3_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -4
for e in 0 6; do
STLT_GEMM_SPLIT_BF16=$e timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/x3/t$e.json
python - <<PY
import json
j=json.loads(open('gpurun_out/x3/t$e.json').read())
print("split=$e", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("launches_per_step"))
PY
done
