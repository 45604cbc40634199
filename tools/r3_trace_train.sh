#!/bin/bash
# rocprofv3 kernel trace of the training bench; prints the timeline summary of one step and keeps the csvs under gpurun_out/<tag>/
tag=${1:-r3_train_trace}; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o o -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 "$@" > $O/bench.log 2>&1
cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
T=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
cp $T $O/kernel_trace.csv
python3 $R/tools/step_trace.py $T > $O/step_timeline.txt
python3 $R/tools/step_trace.py $T --summary
tail -1 $O/bench.log | cut -c1-300
