#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_run6; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn_core_bwd or attention_autograd" > $O/pytest_attn.log 2>&1; tail -3 $O/pytest_attn.log
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_shape_sweep_gpu.py tests/test_caf.py -x -q -m gpu > $O/pytest_train.log 2>&1; tail -3 $O/pytest_train.log
for mode in 1 2; do for b in 16 64; do STLT_ATTN_BWD16=$mode python bench.py --mode train --config cfg4 --batch $b --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/train_cfg4_b${b}_m$mode.json; python - $O/train_cfg4_b${b}_m$mode.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read()); k=j["kernel_ms_per_step"]; print(sys.argv[1].split('/')[-1], j["value"], "clips/s", j["ms_per_step"], "ms gemm", k["gemm"], "attn_bwd", k["attn_bwd"], "frac", j["roofline"]["frac"])
PY
done; done
python tools/bench_caf.py --train --batch 64 2>/dev/null | tail -1 | cut -c1-250
