#!/bin/bash
mkdir -p gpurun_out/x3
timeout 900 python -m pytest tests/test_gemm_bf16x3_gpu.py -x -q 2>&1 | tail -3
echo "== 16x16x32"
timeout 600 python tools/bench_gemm_bf16x3.py > gpurun_out/x3/shapes_m16.txt 2>&1
tail -8 gpurun_out/x3/shapes_m16.txt | head -6 | cut -c1-45,115-
echo "== 32x32x16"
STLT_HIP_LIB=build/variants/libstlt_hip_x3m32.so timeout 600 python tools/bench_gemm_bf16x3.py > gpurun_out/x3/shapes_m32.txt 2>&1
tail -8 gpurun_out/x3/shapes_m32.txt | head -6 | cut -c1-45,115-
echo "== 16x16x32 again"
timeout 600 python tools/bench_gemm_bf16x3.py 2>&1 | tail -8 | head -6 | cut -c1-45,115-
echo "== stamps 16x16x32"
STLT_HIP_LIB=build/variants/libstlt_hip_x3stamp.so timeout 300 python tools/x3_stamps.py 2>&1 | grep -v "wave  [1235679]\|wave 1[01]\|amdgpu.ids"
