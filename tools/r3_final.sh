#!/bin/bash
# final validation of the round: smoke(), the full GPU suite, then the profile collection
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_final; mkdir -p $O; cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash tools/collect_round3.sh > $O/collect.log 2>&1; tail -6 $O/collect.log | cut -c1-250
