#!/bin/bash
# usage: tools/prof_kernels.sh <tag> <python script + args...>   -> prints the kernel-stats rows, copies csv to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o o -- python3 "$@" > /tmp/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then echo "no stats; log tail:"; tail -5 /tmp/prof_$tag.log; exit 1; fi
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}')
PY
