#!/bin/bash
mkdir -p gpurun_out/x3
timeout 900 python -m pytest tests/test_gemm_bf16x3_gpu.py -x -q 2>&1 | tail -3
for v in x3stamp0; do
echo "== $v"
STLT_HIP_LIB=build/variants/libstlt_hip_$v.so timeout 300 python tools/x3_stamps.py 2>&1 | grep -v "wave  [1235679]\|wave 1[0-5]\|amdgpu.ids"
done
timeout 600 python tools/bench_gemm_bf16x3.py > gpurun_out/x3/shapes.txt 2>&1
tail -8 gpurun_out/x3/shapes.txt | cut -c1-45,115-
