#!/bin/bash
mkdir -p gpurun_out/x3
for v in x3e1 x3e2 x3e3; do
  STLT_HIP_LIB=build/variants/libstlt_hip_$v.so timeout 600 python tools/bench_gemm_bf16x3.py > gpurun_out/x3/shapes_$v.txt 2>&1
  echo "== $v"; tail -8 gpurun_out/x3/shapes_$v.txt | head -6 | cut -c1-40,110-150,190-
done
