#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel statistics and the two PMC passes of the default bench command,
# written to gpurun_out/ (copy what should be judged into profiles/).  python3 goes straight after `--`.
R=${GRAFT_REPO_ROOT:-$PWD}
tag=${1:-round2}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rm -rf /tmp/ks /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o o -- python3 $R/bench.py --no-cpu-baseline --no-skip-padding > $R/gpurun_out/${tag}_bench_under_rocprof.log 2>&1
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${tag}_kernel_stats_b1024.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o o -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-padding > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o o -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-padding > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $R/gpurun_out/${tag}_traffic_pmc.json /tmp/pf /tmp/pw
head -12 $R/gpurun_out/${tag}_kernel_stats_b1024.csv
tail -1 $R/gpurun_out/${tag}_bench_under_rocprof.log | cut -c1-200
rm -rf /tmp/pu
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pu -o o -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-padding > /dev/null 2>&1
python3 $R/tools/pmc_util.py $R/gpurun_out/${tag}_util_pmc.json /tmp/pu
