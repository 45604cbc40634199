#!/usr/bin/env python3
"""Register / spill / scratch metadata of every kernel in one of the library's objects (build/obj/<stem>.*.o, newest).

    python tools/kernel_regs.py gemm [name-filter]
"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
stem = sys.argv[1] if len(sys.argv) > 1 else "gemm"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
objs = sorted(glob.glob(os.path.join(ROOT, "build", "obj", stem + ".*.o")), key=os.path.getmtime)
if not objs:
    raise SystemExit("no object for " + stem)
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", objs[-1]], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    if "--keep" in sys.argv:
        import shutil; shutil.copy(co, "/tmp/%s.co" % stem); print("/tmp/%s.co" % stem)
for blk in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    if flt and flt not in dem:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print(f"{dem:90s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} vspill {g('vgpr_spill_count'):>3} sspill {g('sgpr_spill_count'):>3} "
          f"scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6} kernarg {g('kernarg_segment_size'):>5}")
