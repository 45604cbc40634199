"""Build libstlt_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain C-ABI .so."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstlt_hip.so")
SOURCES = ["api.hip", "rowwise.hip", "gemm.hip", "attn.hip", "backward.hip", "train.hip", "collate.hip", "caf.hip", "ragged.hip", "optim.hip", "bwd_api.hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "stlt_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, *extra_flags,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print("[stlt build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
