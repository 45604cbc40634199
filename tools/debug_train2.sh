#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
STLT_BENCH_ONE_GPU=1 STLT_BENCH_FAULT_DUMP=90 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 bench.py --mode train --gpus 2 --steps 2 --warmup 1 --batch 8 > gpurun_out/debug_train2.log 2>&1
echo rc=$?
tail -80 gpurun_out/debug_train2.log
