#!/bin/bash
# full GPU suite + the default bench line (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-r3_full}; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-3000
