"""What the compiler made of the kernels: read from the code objects inside the built library (no GPU needed).

A kernel with a private (scratch) segment has spilled registers.  In this code base that is never harmless: the reloads are
`scratch_load` + `s_waitcnt vmcnt(0)`, and the hot loops keep LDS-DMA / global loads in flight that they count by hand.  Both
times it happened (the attention kernel at a forced 128-register bound; the GEMM with loop-invariant 64-bit lane addresses
hoisted out of its K loop) it cost several per cent end to end without failing a single numerics test."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402

LIB = os.path.join(ROOT, "keep_amd", "libkeep_hip.so")
needs_tools = pytest.mark.skipif(
    not (os.path.exists(LIB) and shutil.which("objcopy") and os.path.exists(os.path.join(KR.LLVM_BIN, "llvm-readelf"))
         and os.path.exists(os.path.join(KR.LLVM_BIN, "clang-offload-bundler"))),
    reason="needs the built library, objcopy and the ROCm llvm tools")


@pytest.fixture(scope="module")
def kernels():
    return KR.kernel_resources(LIB)


@needs_tools
def test_no_kernel_spills(kernels):
    assert len(kernels) > 40
    bad = [(k["name"], k["private_segment_fixed_size"]) for k in kernels if k["private_segment_fixed_size"] or k.get("vgpr_spill_count", 0)]
    assert not bad, f"kernels with a scratch segment (spills): {bad}"


@needs_tools
def test_register_budgets_match_the_planned_occupancy(kernels):
    by = {k["name"]: k for k in kernels}
    # 256x256 GEMM: 8 waves per CU = 2 per SIMD -> 256 registers each (accumulators included)
    gemm = [k for n, k in by.items() if "gemm_f16_v2_kernelILi256ELi2ELi4ELi4E" in n]
    assert len(gemm) >= 10
    assert all(k["vgpr_count"] + k.get("agpr_count", 0) <= 256 for k in gemm)
    # ViT attention (13 key tiles, 8 waves): two workgroups per CU by LDS = 4 waves per SIMD -> 128 registers
    att = by["_ZN5keepk16attention_kernelILi13ELb0ELi8EEEv10AttnParams"]
    assert att["vgpr_count"] <= 128 and att["max_flat_workgroup_size"] == 512
    att16 = by["_ZN5keepk16attention_kernelILi16ELb0ELi8EEEv10AttnParams"]
    assert att16["vgpr_count"] <= 128
