#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_run8; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_train_gpu.py tests/test_shape_sweep_gpu.py -x -q -m gpu > $O/pytest_train.log 2>&1; tail -3 $O/pytest_train.log
for m in 1 2; do STLT_ATTN_BWD16=$m python tools/bench_train.py --skip-padding --dropout 0.1 2>/dev/null | tail -1 | cut -c1-400; done
for m in 1 2; do STLT_ATTN_BWD16=$m python tools/bench_train.py --skip-padding --dropout 0.1 --config cfg4 --batch 16 2>/dev/null | tail -1 | cut -c1-400; done
