#!/bin/bash
# final validation of the round: smoke(), the full GPU suite, then the profile collection
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_final; mkdir -p $O; cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
# the same suite with the opt-in split-bf16 products switched on for the whole process (every parity tolerance unchanged)
STLT_GEMM_SPLIT_BF16=6 timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_split_bf16.log 2>&1; tail -2 $O/pytest_gpu_split_bf16.log
bash tools/collect_round3.sh > $O/collect.log 2>&1; tail -6 $O/collect.log | cut -c1-250
