#!/bin/bash
mkdir -p gpurun_out/x3
timeout 900 python -m pytest tests/test_gemm_bf16x3_gpu.py tests/test_bench_gpu.py -x -q 2>&1 | tail -3
timeout 600 python tools/bench_gemm_bf16x3.py 2>&1 | tail -8 | cut -c1-45,115-
