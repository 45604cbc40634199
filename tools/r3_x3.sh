#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the split-bf16 kernel's own measurements — per-shape table, per-wave shader-clock stamps of the
# -DSTLT_X3_STAMP=1 variant (build it first: build.variant("x3stamp", {"gemm_bf16x3.hip": ["-DSTLT_X3_STAMP=1"]})), and the
# utilisation PMC pass (its own rocprofv3 run, kernel trace only).  Output under gpurun_out/x3/.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/x3; mkdir -p $O; cd $R
timeout 600 python tools/bench_gemm_bf16x3.py > $O/round3_gemm_bf16x3.txt 2>&1
tail -8 $O/round3_gemm_bf16x3.txt | cut -c1-45,115-
if [ -f build/variants/libstlt_hip_x3stamp.so ]; then
  STLT_HIP_LIB=build/variants/libstlt_hip_x3stamp.so timeout 300 python tools/x3_stamps.py > $O/round3_gemm_bf16x3_stamps.txt 2>&1
  grep "TFLOP\|clock\|MFMA waves\|producer waves" $O/round3_gemm_bf16x3_stamps.txt
fi
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/px
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/px -o o -- python3 $R/tools/bench_gemm_bf16x3.py --iters 3 > $O/pmc_run.log 2>&1
python3 $R/tools/pmc_util.py $O/round3_util_pmc_split_bf16.json /tmp/px
python3 - <<PY
import json
j = json.load(open("$O/round3_util_pmc_split_bf16.json"))
for k, v in j["kernels"].items():
    if "gemm" in k: print(k, v)
PY
