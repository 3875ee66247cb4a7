#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in a built libkeep_hip.so, read from the code objects' metadata notes.

    python tools/kernel_resources.py [path/to/libkeep_hip.so]

A kernel with a private segment has spilled registers: its reloads are `scratch_load` + `s_waitcnt vmcnt(0)`, which drains the
LDS-DMA / global-load queue the hot loops count by hand.  tests/test_build_artifacts.py asserts that no product kernel has one.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from typing import Dict, List

LLVM_BIN = os.environ.get("KEEP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def _run(*cmd: str) -> str:
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def demangle(names: List[str]) -> List[str]:
    filt = os.path.join(LLVM_BIN, "llvm-cxxfilt")
    if not names or not os.path.exists(filt):
        return names
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return out if len(out) == len(names) else names


def kernel_resources(lib: str) -> List[Dict]:
    """One dict per kernel: name, vgpr, agpr, sgpr, scratch (bytes per lane), lds (static bytes), spills."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        if not starts:
            raise RuntimeError(f"{lib}: no offload bundle found in .hip_fatbin")
        rows: List[Dict] = []
        for i, a in enumerate(starts):
            b = starts[i + 1] if i + 1 < len(starts) else len(blob)
            chunk, co = os.path.join(td, f"b{i}.bin"), os.path.join(td, f"b{i}.co")
            open(chunk, "wb").write(blob[a:b])
            subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--targets={TARGET}",
                            f"--input={chunk}", f"--output={co}"], check=True, capture_output=True)
            notes = _run(os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co)
            cur: Dict = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'\"")
                if k == "agpr_count" and cur.get("name"):      # first key of a new kernel record in the note (alphabetical order)
                    rows.append(cur); cur = {}
                if k in ("name", "symbol"):
                    cur[k] = v
                elif k in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
                           "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size"):
                    cur[k] = int(v)
            if cur.get("name"):
                rows.append(cur)
    rows = [r for r in rows if "vgpr_count" in r]
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["pretty"] = d
    return rows


def main() -> None:
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "keep_amd", "libkeep_hip.so")
    rows = kernel_resources(lib)
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'spill':>6} {'lds':>7}  kernel")
    for r in sorted(rows, key=lambda r: r["pretty"]):
        print(f"{r['vgpr_count']:5d} {r.get('agpr_count', 0):5d} {r['sgpr_count']:5d} {r['private_segment_fixed_size']:8d} "
              f"{r.get('vgpr_spill_count', 0):6d} {r['group_segment_fixed_size']:7d}  {r['pretty'][:150]}")
    bad = [r for r in rows if r["private_segment_fixed_size"]]
    print(f"{len(rows)} kernels, {len(bad)} with a private segment")


if __name__ == "__main__":
    main()
