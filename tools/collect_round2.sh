#!/bin/bash
# Run ON THE GPU BOX (via gpurun): every measurement profiles/round2_* is made from.  Writes under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/round2_bench_b1024.json 2> $O/round2_bench_b1024.err
python bench.py --mode train > $O/round2_bench_train_b64.json 2> $O/round2_bench_train_b64.err
: > $O/round2_bench_sweep.jsonl
for args in "--config cfg2 --batch 64" "--config cfg2 --batch 256" "--config cfg2 --batch 1024" "--config cfg4 --batch 16" "--config cfg4 --batch 64" "--config cfg1 --batch 4096"; do
  python bench.py $args --no-cpu-baseline 2>/dev/null | tail -1 >> $O/round2_bench_sweep.jsonl
done
python tools/bench_gemm.py --batch 1024 --iters 10 > $O/round2_gemm_shapes_b1024.txt 2>&1
python tools/bench_gemm.py --batch 64 --iters 20 > $O/round2_gemm_shapes_b64.txt 2>&1
python tools/bench_attn.py --batches 64 256 1024 > $O/round2_attn_shapes.txt 2>&1
bash tools/collect_profiles.sh round2 > $O/round2_collect.log 2>&1
bash tools/prof_kernels.sh round2_train_step $R/bench.py --mode train --no-cpu-baseline --steps 10 > $O/round2_train_prof.log 2>&1
# kernel-only durations (rocprofv3 begin/end stamps, no launch gaps) of the small-batch and cfg4 launches
bash tools/prof_kernels.sh round2_cfg2_b64 $R/bench.py --config cfg2 --batch 64 --no-cpu-baseline --no-skip-padding > $O/round2_cfg2_b64_prof.log 2>&1
bash tools/prof_kernels.sh round2_cfg2_b256 $R/bench.py --config cfg2 --batch 256 --no-cpu-baseline --no-skip-padding > $O/round2_cfg2_b256_prof.log 2>&1
bash tools/prof_kernels.sh round2_cfg4_b64 $R/bench.py --config cfg4 --batch 64 --no-cpu-baseline --no-skip-padding > $O/round2_cfg4_b64_prof.log 2>&1
python tools/bench_gemm_train.py > $O/round2_gemm_train_shapes_b64.txt 2>&1
tail -3 $O/round2_collect.log
tail -1 $O/round2_bench_b1024.json | cut -c1-400
