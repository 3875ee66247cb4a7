"""Build libkeep_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m keep_amd.build [--force]

The shared object lands next to this file (keep_amd/libkeep_hip.so); it is git-ignored but
travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libkeep_hip.so")
SOURCES = ["gemm_f16.hip", "gemm_f16_v2.hip", "gemm_f16_skinny.hip", "attention.hip", "rowops.hip", "sgemm_f32.hip", "wsi.hip", "launch_util.hip", "engine.hip"]
# -falign-loops=64: the hot loops start on an instruction-cache line.  Without it a functionally identical edit elsewhere in a kernel moved the
# K loop of the GEMM by a few dwords and the whole encoder by up to 3 % (measured: DESIGN.md section 4); with it +0.5 % and reproducible.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-falign-loops=64", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value",
         "-Wno-unused-result"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


HEADERS = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + [os.path.join(HERE, "..", "include", "keep_hip.h")]
MANIFEST = os.path.join(HERE, "build", "manifest.json")


def _sha(paths, extra="") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _sources(defines):
    return list(SOURCES)


def _source_keys(defines):
    """One content hash per object: the source, every header and the flags (mtimes play no part: the prebuilt .so that
    travels with a snapshot has arbitrary timestamps)."""
    hdr = _sha(HEADERS, " ".join(FLAGS + list(defines)))
    return {s: _sha([os.path.join(CSRC, s)], hdr) for s in _sources(defines)}


def _read_manifest(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def up_to_date() -> bool:
    m = _read_manifest(MANIFEST)
    return os.path.exists(OUT) and m.get("objects") == _source_keys([]) and m.get("lib") == _sha([OUT])


def build(force: bool = False, verbose: bool = True) -> str:
    # experiment builds: KEEP_BUILD_DEFINES="-DKEEP_A_AUX=2" KEEP_BUILD_OUT=libkeep_hip_x.so python -m keep_amd.build
    # (loaded with KEEP_HIP_LIB=<path>, see _lib.py); the default build ignores both
    defines = os.environ.get("KEEP_BUILD_DEFINES", "").split()
    out = os.path.join(HERE, os.environ["KEEP_BUILD_OUT"]) if os.environ.get("KEEP_BUILD_OUT") else OUT
    if defines or out != OUT:
        return _build(out, os.path.join(HERE, "build_" + os.path.basename(out)), defines, verbose)
    if not force and up_to_date():
        if verbose:
            print(f"build: {OUT} matches the content hashes of its {len(SOURCES)} sources: recompiled 0 objects")
        return OUT
    return _build(OUT, os.path.join(HERE, "build"), [], verbose, force)


def _build(OUT: str, objdir: str, defines, verbose: bool, force: bool = False) -> str:
    cc = hipcc()
    os.makedirs(objdir, exist_ok=True)
    manifest_path = os.path.join(objdir, "manifest.json")
    old = {} if force else _read_manifest(manifest_path).get("objects", {})
    keys = _source_keys(defines)
    recompiled = []

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if old.get(src) == keys[src] and os.path.exists(obj):
            return obj
        recompiled.append(src)
        cmd = [cc, *FLAGS, *defines, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    srcs = _sources(defines)
    for f in os.listdir(objdir):                      # objects of sources that have left the tree are not carried along
        if f.endswith(".o") and f.replace(".o", ".hip") not in srcs:
            os.remove(os.path.join(objdir, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, srcs))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(manifest_path, "w") as f:
        json.dump({"objects": keys, "lib": _sha([OUT])}, f, indent=1)
    if verbose:
        print(f"build: recompiled {len(recompiled)} objects ({', '.join(sorted(recompiled)) or 'none'}), linked {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
