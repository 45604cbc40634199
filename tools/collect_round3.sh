#!/bin/bash
# Run ON THE GPU BOX (via gpurun): every measurement profiles/round3_* is made from.  Writes under gpurun_out/round3/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/round3
mkdir -p $O
cd $R
export TMPDIR=/tmp
# 1. the default line (what the driver runs) and the training line
python bench.py > $O/round3_bench_b1024.json 2> $O/round3_bench_b1024.err
python bench.py --mode train > $O/round3_bench_train_b64.json 2> $O/round3_bench_train_b64.err
# 2. batch / config sweep of the forward
: > $O/round3_bench_sweep.jsonl
for args in "--config cfg2 --batch 64" "--config cfg2 --batch 256" "--config cfg4 --batch 16" "--config cfg4 --batch 64" "--config cfg1 --batch 4096"; do
  python bench.py $args --no-cpu-baseline --no-side-legs 2>/dev/null | tail -1 >> $O/round3_bench_sweep.jsonl
done
# 3. per-shape products of the training step, per-shape attention
python tools/bench_gemm_train.py > $O/round3_gemm_train_shapes_b64.txt 2>&1
python tools/bench_gemm.py --batch 1024 --iters 10 > $O/round3_gemm_shapes_b1024.txt 2>&1
python tools/bench_mhsa.py > $O/round3_mhsa_ab.jsonl 2>&1
python tools/bench_gemm_bf16x3.py > $O/round3_gemm_bf16x3.txt 2>&1
if [ -f build/variants/libstlt_hip_x3stamp.so ]; then
  STLT_HIP_LIB=build/variants/libstlt_hip_x3stamp.so python tools/x3_stamps.py > $O/round3_gemm_bf16x3_stamps.txt 2>&1
  STLT_HIP_LIB=build/variants/libstlt_hip_x3stamp.so python tools/x3_stamps.py 229376 2304 768 >> $O/round3_gemm_bf16x3_stamps.txt 2>&1
fi
# 4. rocprofv3: kernel statistics of the default command, then the separate PMC passes (traffic, utilisation)
cd /tmp
rm -rf /tmp/ks /tmp/pf /tmp/pw /tmp/pu
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o o -- python3 $R/bench.py --no-cpu-baseline --no-skip-padding --no-split-bf16 --no-side-legs > $O/round3_bench_under_rocprof_b1024.log 2>&1
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $O/round3_kernel_stats_b1024.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o o -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-padding --no-split-bf16 --no-side-legs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o o -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-padding --no-split-bf16 --no-side-legs > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $O/round3_traffic_pmc.json /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pu -o o -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-padding --no-split-bf16 --no-side-legs > /dev/null 2>&1
python3 $R/tools/pmc_util.py $O/round3_util_pmc.json /tmp/pu
# 4b. the same forward with the opt-in split-bf16 products, under the tracer (kernel statistics only)
rm -rf /tmp/kx
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx -o o -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-skip-padding --split-bf16-main --no-side-legs > $O/round3_bench_split_bf16_under_rocprof_b1024.log 2>&1
cp $(find /tmp/kx -name '*kernel_stats.csv' | head -1) $O/round3_kernel_stats_split_bf16_b1024.csv
# 5. the training step under the tracer: kernel statistics + the timeline of one step
rm -rf /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o o -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 > $O/round3_train_under_rocprof.log 2>&1
cp $(find /tmp/pt -name '*kernel_stats.csv' | head -1) $O/round3_train_step_kernel_stats_b64.csv
python3 $R/tools/step_trace.py $(find /tmp/pt -name '*kernel_trace.csv' | head -1) --summary > $O/round3_train_step_timeline_b64.txt
cd $R
python tools/bench_caf.py > $O/round3_bench_caf.jsonl 2>&1
python tools/bench_caf.py --train --batch 32 >> $O/round3_bench_caf.jsonl 2>&1
python tools/bench_caf.py --train --batch 64 >> $O/round3_bench_caf.jsonl 2>&1
tail -1 $O/round3_bench_b1024.json | cut -c1-300
tail -1 $O/round3_bench_train_b64.json | cut -c1-300
head -8 $O/round3_kernel_stats_b1024.csv | cut -c1-200
