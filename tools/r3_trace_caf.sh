#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_caf_trace; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_caf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_caf -o o -- python3 $R/tools/bench_caf.py --train --batch 64 --steps 4 --warmup 2 > $O/bench.log 2>&1
cp $(find /tmp/prof_caf -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 $R/tools/step_trace.py $(find /tmp/prof_caf -name '*kernel_trace.csv' | head -1) --summary | head -50
tail -1 $O/bench.log | cut -c1-300
