#!/bin/bash
# utilisation counters of the split-bf16 kernel (its own PMC pass, kernel trace only)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/x3; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/px
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/px -o o -- python3 $R/tools/bench_gemm_bf16x3.py --iters 3 > $O/pmc_run.log 2>&1
python3 $R/tools/pmc_util.py $O/round3_util_pmc_split_bf16.json /tmp/px
python3 - <<PY
import json
j=json.load(open("$O/round3_util_pmc_split_bf16.json"))
for k,v in j.items():
    if "gemm" in k: print(k, v)
PY
rm -rf /tmp/py
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT --output-format csv -d /tmp/py -o o -- python3 $R/tools/bench_gemm_bf16x3.py --iters 3 > $O/pmc_run2.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("/tmp/py/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "bf16x3" if "bf16x3" in r["Kernel_Name"] else ("gemm_nt" if "gemm_nt_kernel" in r["Kernel_Name"] else None)
        if not k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k in acc: print(k, n[k], {c: round(v / max(n[k],1)) for c, v in acc[k].items()})
PY
tail -3 $O/pmc_run2.log
