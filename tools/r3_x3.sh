#!/bin/bash
mkdir -p gpurun_out/x3
timeout 900 python -m pytest tests/test_gemm_bf16x3_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --no-side-legs --no-cpu-baseline > gpurun_out/x3/bench.json 2> gpurun_out/x3/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/x3/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"]); print(j.get("split_bf16"))
PY
