#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the per-launch bound tables of the 64-clip regime, with rocprofv3's per-kernel durations.
#   tools/collect_launch_bound.sh [tag]   ->  gpurun_out/r6/<tag>launch_bound_{refdef,cfg2p,cfg2}_b64.txt, ..._train_b64.txt, ..._cacnf_train_b64.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6
TAG=$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
one() {  # name, launch_bound args
  rm -rf /tmp/lb_$1
  rocprofv3 --kernel-trace --output-format csv -d /tmp/lb_$1 -o lb -- python3 $R/tools/launch_bound.py $2 --notes /tmp/lb_$1.json > /tmp/lb_$1.log 2>&1
  python3 $R/tools/launch_bound.py --merge /tmp/lb_$1.json --trace $(find /tmp/lb_$1 -name '*kernel_trace.csv' | head -1) > $O/${TAG}launch_bound_$1.txt 2>&1 || { tail -5 /tmp/lb_$1.log; }
  grep "^# step" $O/${TAG}launch_bound_$1.txt
}
one refdef_b64 "--config refdef --batch 64"
one cfg2p_b64 "--config cfg2p --batch 64"
one cfg2_b64 "--config cfg2 --batch 64"
one train_b64 "--config cfg2 --batch 64 --mode train"
one cacnf_train_b64 "--config cfg2 --batch 64 --mode cacnf_train --steps 10"
one cfg2_b1024 "--config cfg2 --batch 1024 --steps 10 --warmup 3"   # the headline forward, for reference
