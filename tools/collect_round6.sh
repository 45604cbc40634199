#!/bin/bash
# Run ON THE GPU BOX (via gpurun): every end-of-round measurement profiles/round6_* is made from.  Writes under gpurun_out/round6/.
# (the one collection script: earlier rounds' scripts differed in file names only and are in the history)
#   part 1 (default): bench lines, sweeps, shape tables, rocprofv3 kernel statistics + PMC passes
#   part 2 ($1 = suites): soak + the GPU suite under its three switch settings
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/round6
mkdir -p $O
cd $R
export TMPDIR=/tmp
NS="--no-cpu-baseline --no-skip-padding --no-split-bf16 --no-side-legs"
if [ "$1" == "suites" ]; then
  python tools/soak.py --reps 200 > $O/round6_soak.jsonl 2>/dev/null
  python -m pytest tests -q -m gpu > $O/round6_pytest_gpu.log 2>&1
  STLT_GEMM_SPLIT_BF16=6 python -m pytest tests -q -m gpu > $O/round6_pytest_gpu_split_bf16_on.log 2>&1
  STLT_FUSED_MHSA=0 STLT_GEMM16=0 STLT_TRAIN_DW_STREAM=0 STLT_TRAIN_DEFER_REDUCE=0 STLT_ATTN16_TAIL=0 STLT_BLOCK_DW_DEFER=0 STLT_FFN1_KEEP_FUSED=0 STLT_TRAIN_WT=0 STLT_ATTN_BWDX16=0 STLT_ATTN16_DROPOUT=0 python -m pytest tests -q -m gpu > $O/round6_pytest_gpu_dispatches_off.log 2>&1
  for f in $O/round6_pytest_gpu.log $O/round6_pytest_gpu_split_bf16_on.log $O/round6_pytest_gpu_dispatches_off.log; do tail -n 1 $f; done
  python - <<PY
import json
rows = [r for r in (json.loads(l) for l in open("$O/round6_soak.jsonl") if l.startswith("{")) if "case" in r]
print("soak cases", len(rows), "failing", sum(1 for r in rows if r["not_bit_identical"] or r["out_of_tolerance"]))
PY
  exit 0
fi
# 1. the default line (what the driver runs) and the training line
python bench.py > $O/round6_bench_b1024.jsonl 2> $O/round6_bench_b1024.err   # one JSON line per side leg, the contract line last
python bench.py --mode train > $O/round6_bench_train_b64.json 2> $O/round6_bench_train_b64.err
# 2. batch / config sweep of the forward (the reference's real layouts included), and the 64-clip points with the small-tile routing off
: > $O/round6_bench_sweep.jsonl
for args in "--config cfg2 --batch 64" "--config cfg2 --batch 256" "--config cfg2p --batch 64" "--config cfg2p --batch 1024" "--config refdef --batch 64" \
            "--config refdef --batch 1024" "--config cfg4 --batch 16" "--config cfg4 --batch 64" "--config cfg1 --batch 4096"; do
  python bench.py $args $NS 2>/dev/null | tail -1 >> $O/round6_bench_sweep.jsonl
done
: > $O/round6_bench_sweep_large_tiles_only.jsonl
for args in "--config cfg2 --batch 64" "--config cfg2p --batch 64" "--config refdef --batch 64" "--config cfg4 --batch 64"; do
  STLT_GEMM16=0 python bench.py $args $NS 2>/dev/null | tail -1 >> $O/round6_bench_sweep_large_tiles_only.jsonl
done
# 3. per-shape product tables, the fused MHSA sweep, attention cores
python tools/bench_gemm16.py --dx > $O/round6_gemm16_shapes.jsonl 2>&1
python tools/bench_gemm16.py --rows 32768 229376 --iters 8 >> $O/round6_gemm16_shapes.jsonl 2>&1
python tools/bench_gemm_train.py > $O/round6_gemm_train_shapes_b64.txt 2>&1
python tools/bench_gemm.py --batch 1024 --iters 10 > $O/round6_gemm_shapes_b1024.txt 2>&1
python tools/bench_mhsa.py --frames 32 17 33 64 --clips 64 256 1024 --train > $O/round6_mhsa_fused_ab.jsonl 2>&1
python tools/bench_mhsa.py --frames 7 5 8 36 --clips 2048 8192 32768 --noncausal >> $O/round6_mhsa_fused_ab.jsonl 2>&1
python tools/bench_attn.py --batches 64 1024 > $O/round6_attn_shapes.txt 2>&1
# 4. rocprofv3 kernel statistics: the default command, the reference's real layouts at 1024 clips, and the 64-clip operating points
cd /tmp
stats() {  # tag, bench args
  rm -rf /tmp/ks_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$1 -o o -- python3 $R/bench.py $2 $NS > $O/round6_bench_under_rocprof_$1.log 2>&1
  cp $(find /tmp/ks_$1 -name '*kernel_stats.csv' | head -1) $O/round6_kernel_stats_$1.csv
}
stats b1024 ""
stats cfg2p_b1024 "--config cfg2p --batch 1024 --steps 10 --warmup 3"
stats refdef_b1024 "--config refdef --batch 1024 --steps 10 --warmup 3"
stats cfg2_b64 "--config cfg2 --batch 64"
stats cfg2p_b64 "--config cfg2p --batch 64"
stats refdef_b64 "--config refdef --batch 64"
stats cfg4_b64 "--config cfg4 --batch 64"
# 5. separate PMC passes of the default command (traffic, utilisation)
rm -rf /tmp/pf /tmp/pw /tmp/pu
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o o -- python3 $R/bench.py --steps 2 --warmup 1 $NS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o o -- python3 $R/bench.py --steps 2 --warmup 1 $NS > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $O/round6_traffic_pmc.json /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pu -o o -- python3 $R/bench.py --steps 2 --warmup 1 $NS > /dev/null 2>&1
python3 $R/tools/pmc_util.py $O/round6_util_pmc.json /tmp/pu
# 5b. the same two traffic passes for the released checkpoints' layout (T = 33: the temporal attention core runs as its own kernel there)
rm -rf /tmp/pf2 /tmp/pw2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf2 -o o -- python3 $R/bench.py --config cfg2p --batch 1024 --steps 2 --warmup 1 $NS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw2 -o o -- python3 $R/bench.py --config cfg2p --batch 1024 --steps 2 --warmup 1 $NS > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $O/round6_traffic_pmc_cfg2p.json /tmp/pf2 /tmp/pw2
# 6. the training step under the tracer: kernel statistics + the timeline of one step (two streams = default; one stream beside it)
for mode in 1 0; do
  rm -rf /tmp/pt$mode
  STLT_TRAIN_DW_STREAM=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt$mode -o o -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 > $O/round6_train_under_rocprof_dw$mode.log 2>&1
  python3 $R/tools/step_trace.py $(find /tmp/pt$mode -name '*kernel_trace.csv' | head -1) --summary --back 6 > $O/round6_train_step_timeline_b64_dw$mode.txt
done
cp $(find /tmp/pt1 -name '*kernel_stats.csv' | head -1) $O/round6_train_step_kernel_stats_b64.csv
cd $R
# 7. fusion models
python tools/bench_caf.py > $O/round6_bench_caf.jsonl 2>&1
python tools/bench_caf.py --train --batch 32 >> $O/round6_bench_caf.jsonl 2>&1
python tools/bench_caf.py --train --batch 64 >> $O/round6_bench_caf.jsonl 2>&1
rm -rf /tmp/pc
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o o -- python3 $R/tools/bench_caf.py --train --batch 64 --steps 12 --warmup 3 > /dev/null 2>&1
python3 $R/tools/step_trace.py $(find /tmp/pc -name '*kernel_trace.csv' | head -1) --marker sumsq_kernel --summary > $O/round6_caf_train_step_timeline_b64.txt
cd $R
# the switches of the fusion-model step, one box: block weight-gradient deferral, FFN1 keep epilogue, cross-attention backward, attn16 dropout
: > $O/round6_caf_switches_ab.jsonl
for sw in "" "STLT_BLOCK_DW_DEFER=0" "STLT_FFN1_KEEP_FUSED=0" "STLT_ATTN_BWDX16=0" "STLT_ATTN16_DROPOUT=0" "STLT_TRAIN_WT=0" ""; do
  echo "{\"switch\": \"${sw:-default}\"}" >> $O/round6_caf_switches_ab.jsonl
  env $sw python tools/bench_caf.py --train --batch 64 --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/round6_caf_switches_ab.jsonl
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/round6_smoke.log 2>&1
# 8. round 6: every launch against its bound (rocprofv3 two-pass), the launch floor, the uninitialised-memory probe
tools/collect_launch_bound.sh round6_ > /dev/null 2>&1
for f in $R/gpurun_out/r6/round6_launch_bound_*.txt; do cp $f $O/; done
python tools/launch_floor.py 2>/dev/null | tail -1 > $O/round6_launch_floor.json
python tools/garbage_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/round6_garbage_probe.txt
tail -1 $O/round6_bench_b1024.jsonl | cut -c1-300
tail -1 $O/round6_bench_train_b64.json | cut -c1-300
head -8 $O/round6_kernel_stats_b1024.csv | cut -c1-200
