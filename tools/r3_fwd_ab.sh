#!/bin/bash
# round 3, headline forward A/B: residual add in the out-proj / FFN2 epilogues (with and without the prefetch of the residual
# pieces during a tile's last k-step) and non-temporal epilogue stores.  cfg2, 1024 clips, interleaved repeats.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3_fwd_ab; mkdir -p $O; cd $R
run() { # tag lib fuse
  STLT_HIP_LIB=$R/build/variants/libstlt_hip_$2.so STLT_FUSE_RESIDUAL=$3 python bench.py --no-cpu-baseline --no-skip-padding --no-side-legs --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/$1.json
  python - $O/$1.json $1 <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
k = j["kernel_ms_per_step"]
print(f'{sys.argv[2]:14s} {j["value"]:9.1f} clips/s {j["ms_per_step"]:8.3f} ms  gemm {k["gemm"]:7.3f} ({j["roofline"]["frac"]:.4f})  add_ln {k["add_layernorm"]:.3f}  mhsa {k.get("mhsa_fused", 0):.3f}', flush=True)
PY
}
for rep in 1 2 3; do
  run base_r$rep base 0
  run fuse_r$rep base 1
  run fusepf_r$rep pf 1
  run nt_r$rep nt 0
  run fusepfnt_r$rep pfnt 1
done 2>&1 | tee $O/summary.txt
