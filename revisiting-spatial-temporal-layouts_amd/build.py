"""Build libstlt_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain C-ABI .so.

Each .hip file is compiled to an object under build/obj/ (re-used while the source, the headers and the flags are
unchanged; files compile in parallel) and the objects are linked into <pkg>/libstlt_hip.so.  `variant()` builds the
same library with extra -D flags into build/variants/ for A/B measurements (tools/gemm_ab.py).
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstlt_hip.so")
OBJ = os.path.join(ROOT, "build", "obj")
SOURCES = ["api.hip", "rowwise.hip", "gemm.hip", "attn.hip", "backward.hip", "train.hip", "collate.hip", "caf.hip", "ragged.hip",
           "optim.hip", "bwd_api.hip", "evalk.hip", "attn16.hip", "attn_bwd16.hip", "attn_bwdx16.hip", "wt_cache.hip", "mhsa.hip", "blocks.hip", "gemm_bf16x3.hip", "gemm16.hip", "gemm16_rb4.hip", "gemm16_rb2.hip", "attn_any.hip", "gemm_any.hip"]
BASE_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
# per-source flags of the default build.  gemm_bf16x3.hip: the SLP vectoriser would pair the operand cut's f32 subtractions into
# v_pk_add_f32, which issues slower than two v_sub_f32 beside MFMAs (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever")
FILE_FLAGS = {"gemm_bf16x3.hip": ("-fno-slp-vectorize",)}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _headers():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [os.path.join(ROOT, "include", "stlt_hip.h")]


def _sources():
    missing = [s for s in SOURCES if not os.path.exists(os.path.join(CSRC, s))]
    if missing:
        raise RuntimeError(f"HIP sources missing from {CSRC}: {missing}")
    return list(SOURCES)


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "stlt_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _digest(src: str, flags) -> str:
    h = hashlib.sha1()
    for path in [src, *_headers()]:
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()[:16]


def _compile_one(src_name: str, flags, verbose: bool) -> str:
    src = os.path.join(CSRC, src_name)
    tag = _digest(src, flags)
    obj = os.path.join(OBJ, f"{os.path.splitext(src_name)[0]}.{tag}.o")
    if not os.path.exists(obj):
        os.makedirs(OBJ, exist_ok=True)
        cmd = [_hipcc(), *BASE_FLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC, *flags, "-c", src, "-o", obj + f".{os.getpid()}.tmp"]
        if verbose:
            print("[stlt build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(obj + f".{os.getpid()}.tmp", obj)
        stem = os.path.splitext(src_name)[0] + "."
        mine = sorted((f for f in os.listdir(OBJ) if f.startswith(stem) and f.endswith(".o")), key=lambda f: os.path.getmtime(os.path.join(OBJ, f)))
        for f in mine[:-6]:  # keep the cache small: the six newest objects per source
            os.remove(os.path.join(OBJ, f))
    return obj


def _link(objs, out: str, verbose: bool) -> str:
    # the dynamic symbol table holds the C-ABI only (csrc/exports.map: `stlt_*` — the launchers shared between the sources have
    # C++ linkage and mangled names, which the pattern does not match; tests/test_host_cpu.py compares the table with the header)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", f"-Wl,--version-script={os.path.join(CSRC, 'exports.map')}",
           *objs, "-o", out + f".{os.getpid()}.tmp"]
    if verbose:
        print("[stlt build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + f".{os.getpid()}.tmp", out)
    return out


def _build_to(out: str, per_file_flags, verbose: bool) -> str:
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, tuple(FILE_FLAGS.get(s, ())) + tuple(per_file_flags.get(s, ())) + tuple(per_file_flags.get("*", ())), verbose), srcs))
    return _link(objs, out, verbose)


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB
    return _build_to(LIB, {"*": tuple(extra_flags)}, verbose)


def variant(tag: str, flags_by_file, verbose: bool = False) -> str:
    """Same library with extra flags for some files ({"gemm.hip": ["-DSTLT_GEMM_STAGGER=0"]}) -> build/variants/libstlt_hip_<tag>.so"""
    d = os.path.join(ROOT, "build", "variants")
    os.makedirs(d, exist_ok=True)
    return _build_to(os.path.join(d, f"libstlt_hip_{tag}.so"), dict(flags_by_file), verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
