#!/bin/bash
timeout 900 python -m pytest tests/test_gemm_bf16x3_gpu.py -x -q 2>&1 | tail -8
