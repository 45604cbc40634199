"""tools/launch_bound.py on the CPU: the bound arithmetic (what a judge recomputes by hand) and the merge of a rocprofv3 kernel trace with the
recorder's notes — kernels the runtime / torch start inside a step go to the gaps, a kernel-count mismatch is refused with the names."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

from conftest import ROOT

spec = importlib.util.spec_from_file_location("launch_bound", os.path.join(ROOT, "tools", "launch_bound.py"))
lb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lb)


def test_bounds_follow_the_stated_formula():
    # 1088 x 768 x 768: the best tile is 32 x 128 (204 tiles, one round): 24 k-steps x 2 * 32 * 128 * 32 / (0.94 * 157.3e12 / 256) + epilogue + 1.5 us
    us, how = lb.product_bound(1088, 768, 768)
    t = 2 * 64 * 64 * 32 / (0.94 * 157.3e12 / 256) * 1e6
    assert how.endswith("r1") and abs(us - min(24 * t + 64 * 64 * 4 / (6.3e12 / 256) * 1e6 + 1.5,
                                               24 * 2 * t + 32 * 128 * 4 / (6.3e12 / 256) * 1e6 + 1.5)) < 1e-6
    assert abs(lb.t_kstep_us(128, 192) - 2.72) < 0.01  # the figure the small-tile cost table has for that tile
    b, _ = lb.bound_of({"note": "add_ln rows=1088 d=768", "bytes": 1088 * 768 * 8.0, "flops": 0.0})
    assert abs(b - (1088 * 768 * 8 / 6.3e12 * 1e6 + 1.5)) < 1e-9
    b, _ = lb.bound_of({"note": "", "bytes": 0.0, "flops": 0.0})
    assert b == 1.5
    b, how = lb.bound_of({"note": "gemm16 M=5440 N=2304 K=768 act=0 tile=64x256 tiles=765 wg=256 rounds=3 ksteps=24", "bytes": 0.0, "flops": 1.0})
    assert 120 < b < 140 and how  # 3 rounds of 64 x 256 tiles would be 3 * 24 * 1.82 = 131 us: the bound may pick a better tile, never a worse one


def test_merge_charges_each_launch_its_kernels_and_the_gap_behind_them(tmp_path):
    rows = [[0, "gemm", 0.0, 10.0, "64x64 r1", "gemm M=64 N=768 K=768 act=0 tile=256x128 tiles=6 stream-K wg=36 ksteps/wg=4 (+fix-up)", 1.0, 2],
            [1, "add_layernorm", 0.0, 2.0, "hbm", "add_ln rows=64 d=768", 0.0, 1]]
    notes = tmp_path / "notes.json"
    notes.write_text(json.dumps({"head": "# synthetic", "steps": 2, "wall_us": 30.0, "traced_wall_us": 31.0, "rows": rows}))
    trace = tmp_path / "trace.csv"
    t = 1_000_000
    disp = [("warmup_kernel", t, t + 500), ("void at::native::flip_kernel", t + 1000, t + 1500)]
    for step in range(2):
        base = t + 10_000 + step * 30_000
        disp += [("gemm_nt_kernel<0>", base, base + 8_000), ("gemm_fixup_kernel<0>", base + 9_000, base + 10_000),
                 ("__amd_rocclr_fillBufferAligned", base + 10_500, base + 11_000),  # the runtime's memset: in the gap, not a launch of the library
                 ("add_ln_kernel<3>", base + 12_000, base + 15_000)]
    with open(trace, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for name, s, e in disp:
            w.writerow([name, s, e])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch_bound.py"), "--merge", str(notes), "--trace", str(trace)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip() and l.strip()[0].isdigit()]
    gemm, ln = lines[0].split(), lines[1].split()
    assert abs(float(gemm[2]) - 12.0) < 1e-6 and "[kernel 9.00 us]" in lines[0]  # two kernels (8 + 1 us) and everything up to the next launch's first kernel
    assert abs(float(ln[2]) - 10.5) < 1e-6  # 3 us of kernel + the gap to the next step's first kernel (step 0: 18 us; the last step: none) averaged
    assert "1.0 per step" in r.stdout  # the foreign kernel is reported, not counted as a launch
    # a launcher that starts a kernel the recorder does not count: refused, with the names
    rows[1][7] = 2
    notes.write_text(json.dumps({"head": "# synthetic", "steps": 2, "wall_us": 30.0, "traced_wall_us": 31.0, "rows": rows}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch_bound.py"), "--merge", str(notes), "--trace", str(trace)], capture_output=True, text=True)
    assert r.returncode != 0 and "add_ln_kernel" in (r.stdout + r.stderr)
