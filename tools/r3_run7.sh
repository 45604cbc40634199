#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_run7; mkdir -p $O; cd $R
for f in 0.9 0.99 0.9 0.99 0.95; do STLT_GEMM_SK_FILL=$f python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 > $O/t.json; python - $O/t.json $f <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read()); k=j["kernel_ms_per_step"]; print("sk_fill", sys.argv[2], j["ms_per_step"], "ms gemm", k["gemm"], "frac", j["roofline"]["frac"], "launches", j["roofline"]["launches_per_step"])
PY
done
for f in 0.9 0.99; do STLT_GEMM_SK_FILL=$f python bench.py --batch 64 --no-cpu-baseline --no-skip-padding --steps 30 2>/dev/null | tail -1 > $O/t.json; python - $O/t.json $f <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read()); print("fwd64 sk_fill", sys.argv[2], j["ms_per_step"], "ms", j["roofline"]["frac"])
PY
done
