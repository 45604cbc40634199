#!/bin/bash
timeout 2400 python -m pytest tests/test_gemm_bf16x3_gpu.py tests/test_caf.py tests/test_train_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4
