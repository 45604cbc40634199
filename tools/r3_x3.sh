#!/bin/bash
mkdir -p gpurun_out/x3
timeout 900 python -m pytest tests/test_gemm_bf16x3_gpu.py -x -q 2>&1 | tail -2
echo "== reload late"
timeout 600 python tools/bench_gemm_bf16x3.py 2>&1 | tail -8 | head -6 | cut -c1-45,115-
echo "== reload interleaved"
STLT_HIP_LIB=build/variants/libstlt_hip_x3rl0.so timeout 600 python tools/bench_gemm_bf16x3.py 2>&1 | tail -8 | head -6 | cut -c1-45,115-
echo "== stamps reload late"
STLT_HIP_LIB=build/variants/libstlt_hip_x3stamp.so timeout 300 python tools/x3_stamps.py 2>&1 | grep "TFLOP\|clock\|MFMA waves\|producer waves"
