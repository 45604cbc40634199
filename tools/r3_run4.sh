#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_run4; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -x -q -m gpu --deselect tests/test_bench_gpu.py > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for f in 1 0; do STLT_FUSE_GELU_BWD=$f python bench.py --mode train --no-cpu-baseline > $O/train_gelufuse$f.json 2>/dev/null; python - $O/train_gelufuse$f.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=j["kernel_ms_per_step"]
print(sys.argv[1].split('/')[-1], j["ms_per_step"], "ms", j["value"], "clips/s gemm", k["gemm"], "gelu", k["gelu"], "frac", j["roofline"]["frac"], "launches", j["roofline"]["launches_per_step"])
PY
done
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q -m gpu > $O/pytest_bench.log 2>&1; tail -3 $O/pytest_bench.log
