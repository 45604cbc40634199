#!/bin/bash
for v in x3slp x3stamp0; do
echo "== $v"
STLT_HIP_LIB=build/variants/libstlt_hip_$v.so timeout 300 python tools/x3_stamps.py 2>&1 | grep -v "wave  [1235679]\|wave 1[01]\|amdgpu.ids"
done
