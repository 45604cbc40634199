#!/bin/bash
# A/B of the 16-row attention kernel's launch policy: tools/bench_attn.py per value of STLT_ATTN16_SPLIT_BELOW
# (fraction of the device's wave slots below which a launch is cut into (item, query block) units; 0 = never)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@"; do
  echo "=== STLT_ATTN16_SPLIT_BELOW=$v (pass $rep)"
  STLT_ATTN16_SPLIT_BELOW=$v timeout 300 python tools/bench_attn.py --batches 16 64 128 256 1024 --iters 30
done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_ab.log
