#!/bin/bash
# Run ON THE GPU BOX (via gpurun): every end-of-round measurement profiles/round4_* is made from.  Writes under gpurun_out/round4/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/round4
mkdir -p $O
cd $R
export TMPDIR=/tmp
NS="--no-cpu-baseline --no-skip-padding --no-split-bf16 --no-side-legs"
# 1. the default line (what the driver runs) and the training line
python bench.py > $O/round4_bench_b1024.json 2> $O/round4_bench_b1024.err
python bench.py --mode train > $O/round4_bench_train_b64.json 2> $O/round4_bench_train_b64.err
# 2. batch / config sweep of the forward (the reference's real layouts included), each with the small-tile routing off as well
: > $O/round4_bench_sweep.jsonl
for args in "--config cfg2 --batch 64" "--config cfg2 --batch 256" "--config cfg2p --batch 64" "--config cfg2p --batch 1024" "--config refdef --batch 64" \
            "--config refdef --batch 1024" "--config cfg4 --batch 16" "--config cfg4 --batch 64" "--config cfg1 --batch 4096"; do
  python bench.py $args $NS 2>/dev/null | tail -1 >> $O/round4_bench_sweep.jsonl
done
: > $O/round4_bench_sweep_large_tiles_only.jsonl
for args in "--config cfg2 --batch 64" "--config cfg2p --batch 64" "--config refdef --batch 64" "--config cfg4 --batch 64"; do
  STLT_GEMM16=0 python bench.py $args $NS 2>/dev/null | tail -1 >> $O/round4_bench_sweep_large_tiles_only.jsonl
done
# 3. per-shape product tables: small tiles against large tiles / stream-K (forward and input gradient), training shapes, bench shapes
python tools/bench_gemm16.py --dx > $O/round4_gemm16_shapes.jsonl 2>&1
python tools/bench_gemm_train.py > $O/round4_gemm_train_shapes_b64.txt 2>&1
python tools/bench_gemm.py --batch 1024 --iters 10 > $O/round4_gemm_shapes_b1024.txt 2>&1
# 4. fused MHSA: sequence-length sweep against the pair (the pair with the small-tile routing on, as the dispatch sees it)
python tools/bench_mhsa.py --frames 32 17 33 64 --clips 64 256 1024 --train > $O/round4_mhsa_ab_final.jsonl 2>&1
python tools/bench_mhsa.py --frames 7 5 8 36 --clips 2048 8192 32768 --noncausal >> $O/round4_mhsa_ab_final.jsonl 2>&1
# 5. per-wave stamps of the large-tile GEMM: plain epilogue and the residual-add instantiation (out-proj / FFN2 of a post-norm layer)
for shape in "229376 768 768" "229376 768 3072"; do
  STLT_GEMM_STAMP=1 python tools/gemm_block_times.py $shape >> $O/round4_gemm_wave_stamps.txt 2>&1
  STLT_GEMM_STAMP=1 python tools/gemm_block_times.py $shape --residual >> $O/round4_gemm_wave_stamps.txt 2>&1
done
# 6. rocprofv3: kernel statistics of the default command, then the separate PMC passes (traffic, utilisation)
cd /tmp
rm -rf /tmp/ks /tmp/pf /tmp/pw /tmp/pu
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o o -- python3 $R/bench.py $NS > $O/round4_bench_under_rocprof_b1024.log 2>&1
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $O/round4_kernel_stats_b1024.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o o -- python3 $R/bench.py --steps 2 --warmup 1 $NS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o o -- python3 $R/bench.py --steps 2 --warmup 1 $NS > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $O/round4_traffic_pmc.json /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pu -o o -- python3 $R/bench.py --steps 2 --warmup 1 $NS > /dev/null 2>&1
python3 $R/tools/pmc_util.py $O/round4_util_pmc.json /tmp/pu
# 7. the training step under the tracer: kernel statistics + the timeline of one step (two streams = default; one stream beside it)
for mode in 1 0; do
  rm -rf /tmp/pt$mode
  STLT_TRAIN_DW_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt$mode -o o -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 > $O/round4_train_under_rocprof_dw$mode.log 2>&1
  # --back 6: the last TIMED step (the run ends with 6 event-timed replays that keep one stream whatever the switch says)
  python3 $R/tools/step_trace.py $(find /tmp/pt$mode -name '*kernel_trace.csv' | head -1) --summary --back 6 > $O/round4_train_step_timeline_b64_dw$mode.txt
done
cp $(find /tmp/pt1 -name '*kernel_stats.csv' | head -1) $O/round4_train_step_kernel_stats_b64.csv
# 8. the 64-clip forward and cfg4 under the tracer (kernel-only durations of the small-batch paths)
rm -rf /tmp/k64 /tmp/k4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k64 -o o -- python3 $R/bench.py --batch 64 $NS > /dev/null 2>&1
cp $(find /tmp/k64 -name '*kernel_stats.csv' | head -1) $O/round4_cfg2_b64_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -o o -- python3 $R/bench.py --config cfg4 --batch 64 $NS > /dev/null 2>&1
cp $(find /tmp/k4 -name '*kernel_stats.csv' | head -1) $O/round4_cfg4_b64_kernel_stats.csv
cd $R
# 9. fusion models
python tools/bench_caf.py > $O/round4_bench_caf.jsonl 2>&1
python tools/bench_caf.py --train --batch 32 >> $O/round4_bench_caf.jsonl 2>&1
python tools/bench_caf.py --train --batch 64 >> $O/round4_bench_caf.jsonl 2>&1
STLT_TRAIN_DW_STREAM=0 python tools/bench_caf.py --train --batch 64 >> $O/round4_bench_caf.jsonl 2>&1
STLT_GEMM16=0 python tools/bench_caf.py --train --batch 64 >> $O/round4_bench_caf.jsonl 2>&1
rm -rf /tmp/pc
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o o -- python3 $R/tools/bench_caf.py --train --batch 64 --steps 4 --warmup 2 > /dev/null 2>&1
python3 $R/tools/step_trace.py $(find /tmp/pc -name '*kernel_trace.csv' | head -1) --marker sumsq_kernel --summary > $O/round4_caf_train_step_timeline_b64.txt
cd $R
# 9b. soak: 400 launches of every synchronisation-heavy kernel case, bit-identical and within tolerance
python tools/soak.py --reps 400 > $O/round4_soak.jsonl 2>/dev/null
# 10. the GPU suite on this box: default, with the opt-in split-bf16 products exported, and with the round-4 dispatches off
python -m pytest tests -q -m gpu > $O/round4_pytest_gpu.log 2>&1
STLT_GEMM_SPLIT_BF16=6 python -m pytest tests -q -m gpu > $O/round4_pytest_gpu_split_bf16_on.log 2>&1
STLT_FUSED_MHSA=0 STLT_GEMM16=0 STLT_TRAIN_DW_STREAM=0 STLT_TRAIN_DEFER_REDUCE=0 python -m pytest tests -q -m gpu > $O/round4_pytest_gpu_fused_off.log 2>&1
for f in $O/round4_pytest_gpu.log $O/round4_pytest_gpu_split_bf16_on.log $O/round4_pytest_gpu_fused_off.log; do tail -n 1 $f; done
tail -1 $O/round4_bench_b1024.json | cut -c1-300
tail -1 $O/round4_bench_train_b64.json | cut -c1-300
head -8 $O/round4_kernel_stats_b1024.csv | cut -c1-200
