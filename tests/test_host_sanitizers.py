"""The host halves of every launcher under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY §5 "race detection / sanitizers"; GPU
sanitizers are not available on the pool, and what runs on the host is exactly what they would not see anyway: ~3 k lines of plan
arithmetic — stream-K and grouped-launch plans, the small-tile cost tables, workspace / tape / scratch layouts, overflow guards, argument
checks).  The library's sources are compiled `--offload-host-only` with `-fsanitize=address,undefined -fno-sanitize-recover`, linked with
tests/host_fuzz.hip (the driver) and tests/host_fuzz_stubs.cpp (a stand-in HIP runtime that checks every launch configuration and then
"succeeds", so each entry point runs its whole host-side launch sequence), and the driver calls every C-ABI entry point with random,
tile-boundary and extreme shapes and fake device pointers.  No GPU is involved: it runs in the build container."""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "revisiting-spatial-temporal-layouts_amd", "csrc")
OUT = os.path.join(ROOT, "build", "host_san")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
NM = shutil.which("nm") or "/usr/bin/nm"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-O1", "-g"]
ITERS = 1500


def _digest(paths):
    h = hashlib.sha1(" ".join(SAN).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_host_fuzz():
    """-> path of the sanitized driver (rebuilt when a source, a header or the driver changed)."""
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", "stlt_hip.h"), os.path.join(ROOT, "tests", "host_fuzz.hip"), os.path.join(ROOT, "tests", "host_fuzz_stubs.cpp")]
    tag = _digest(deps)
    exe = os.path.join(OUT, f"host_fuzz.{tag}")
    if os.path.exists(exe):
        return exe
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    flags = ["--offload-host-only", *SAN, "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC]

    def cc(src):
        obj = os.path.join(OUT, os.path.splitext(os.path.basename(src))[0] + ".o")
        r = subprocess.run([HIPCC, *flags, "-c", src, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, srcs + [os.path.join(ROOT, "tests", "host_fuzz.hip")]))
    # every host object references the fat binary its (absent) device pass would have embedded, under a per-file name
    und = subprocess.run([NM, "-u", *objs], capture_output=True, text=True, check=True).stdout
    names = sorted({ln.split()[-1] for ln in und.splitlines() if "__hip_fatbin_" in ln})
    fat = os.path.join(OUT, "fatbin_stubs.c")
    with open(fat, "w") as f:
        f.write("".join(f"char {n}[64];\n" for n in names))
    fat_o = os.path.join(OUT, "fatbin_stubs.o")
    subprocess.run([CLANG.replace("clang++", "clang"), "-c", fat, "-o", fat_o], check=True)
    stubs_o = os.path.join(OUT, "host_fuzz_stubs.o")
    subprocess.run([CLANG, *SAN, "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-c", os.path.join(ROOT, "tests", "host_fuzz_stubs.cpp"), "-o", stubs_o],
                   check=True)
    r = subprocess.run([HIPCC, "-fsanitize=address,undefined", *objs, fat_o, stubs_o, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
@pytest.mark.parametrize("seed", [1, 20261004])
def test_host_launchers_are_clean_under_asan_and_ubsan(seed):
    exe = build_host_fuzz()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")  # the host side is what is under test, wherever this runs
    r = subprocess.run([exe, str(ITERS), str(seed)], capture_output=True, text=True, timeout=900, env=env)
    tail = (r.stdout[-1500:] + "\n" + r.stderr[-6000:])
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
    assert f"host_fuzz: {ITERS} iterations" in r.stdout, tail
    launches = int(r.stdout.split("kernel launches configured: ")[1].split()[0])
    assert launches > 20 * ITERS, tail  # the whole-path entry points really walked their launch sequences
