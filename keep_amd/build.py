"""Build libkeep_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m keep_amd.build [--force]

The shared object lands next to this file (keep_amd/libkeep_hip.so); it is git-ignored but
travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libkeep_hip.so")
SOURCES = ["gemm_f16.hip", "gemm_f16_v2.hip", "gemm_f16_v3.hip", "gemm_f16_skinny.hip", "attention.hip", "rowops.hip", "sgemm_f32.hip", "wsi.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _deps():
    files = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_epilogue.h"),
             os.path.join(HERE, "..", "include", "keep_hip.h")]
    return max(os.path.getmtime(f) for f in files)


def up_to_date() -> bool:
    return os.path.exists(OUT) and os.path.getmtime(OUT) >= _deps()


def build(force: bool = False, verbose: bool = True) -> str:
    # experiment builds: KEEP_BUILD_DEFINES="-DKEEP_A_AUX=2" KEEP_BUILD_OUT=libkeep_hip_x.so python -m keep_amd.build
    # (loaded with KEEP_HIP_LIB=<path>, see _lib.py); the default build ignores both
    defines = os.environ.get("KEEP_BUILD_DEFINES", "").split()
    out = os.path.join(HERE, os.environ["KEEP_BUILD_OUT"]) if os.environ.get("KEEP_BUILD_OUT") else OUT
    if defines or out != OUT:
        return _build(out, os.path.join(HERE, "build_" + os.path.basename(out)), defines, verbose)
    if not force and up_to_date():
        return OUT
    return _build(OUT, os.path.join(HERE, "build"), [], verbose)


def _build(OUT: str, objdir: str, defines, verbose: bool) -> str:
    cc = hipcc()
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, *defines, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
