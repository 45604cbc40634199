#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_run5; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_caf.py tests/test_train_gpu.py tests/test_train_ddp_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for b in 32 64; do python tools/bench_caf.py --train --batch $b 2>/dev/null | tail -1 | cut -c1-300; done
python tools/bench_caf.py --train --batch 64 --stock-loop 2>/dev/null | tail -1 | cut -c1-300
