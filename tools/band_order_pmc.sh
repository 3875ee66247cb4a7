#!/bin/bash
# Run ON THE GPU BOX: what does the 2x over-fetch of the persistent GEMMs cost in clock?  The tile walk of gemm_f16_v2 keeps a band of KEEP_BAND_COLS output columns
# per XCD (2048 in the product: 8 n-tiles share an A panel).  Experiment builds with 1024 (W band of 2 MiB stays in the XCD's L2) and 4096 (an A band stays resident
# across ALL 16 column tiles of fc1) are compared on one box: FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md), microseconds and effective clock per launch of the
# plain fp16 qkv / fc1 / fc2 kernels, one internal stream, 256-tile launches.
#   KEEP_BUILD_DEFINES=-DKEEP_BAND_COLS=1024 KEEP_BUILD_OUT=libkeep_hip_band1024.so python -m keep_amd.build   (build container; the .so travels)
#   gpurun -- 'tools/band_order_pmc.sh libkeep_hip.so libkeep_hip_band1024.so libkeep_hip_band4096.so'  ->  gpurun_out/band_order/*.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/band_order; mkdir -p "$OUT"
export KEEP_CALIBRATE=0
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-sustained --no-breakdown --opt streams=1 --precision fp16"
for lib in "$@"; do
  export KEEP_HIP_LIB=$REPO/keep_amd/$lib
  rm -rf /tmp/bo_f /tmp/bo_c
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/bo_f --output-format csv -- python "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/$lib.fetch.log"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/bo_c --output-format csv -- python "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/$lib.clock.log"
  python - "$(find /tmp/bo_f -name '*counter_collection.csv' | head -1)" "$(find /tmp/bo_c -name '*counter_collection.csv' | head -1)" "$lib" <<'P' | tee "$OUT/$lib.txt"
import csv, sys
from collections import defaultdict
names = {"<256, 2, 4, 4, 0, 0, true>": "qkv  (bias -> fp16)", "<256, 2, 4, 4, 1, 0, true>": "fc1  (bias + GELU -> fp16)", "<256, 2, 4, 4, 2, 0, true>": "fc2  (LayerScale + residual RMW)"}
def tag(k):
    for sig, n in names.items():
        if "gemm_f16_v2_kernel" + sig in k: return n
fetch, clk = defaultdict(list), defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    t = tag(r["Kernel_Name"])
    if t and r["Counter_Name"] == "FETCH_SIZE": fetch[t].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(sys.argv[2])):
    t = tag(r["Kernel_Name"])
    if not t: continue
    clk[t][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": clk[t]["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("library:", sys.argv[3])
for t in names.values():
    if not fetch[t]: continue
    f = sum(fetch[t]) / len(fetch[t]) * 2 * 1024            # FETCH_SIZE is in KiB and counts half the bytes on gfx950 (guide's correction)
    g = sum(clk[t]["GRBM_GUI_ACTIVE"]) / len(clk[t]["GRBM_GUI_ACTIVE"]); m = sum(clk[t]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(clk[t]["SQ_VALU_MFMA_BUSY_CYCLES"])
    d = sum(clk[t]["dur"]) / len(clk[t]["dur"])
    print(f"  {t:36s} launches {len(fetch[t]):4d}  fetched {f / 1e6:8.1f} MB / launch   {d / 1e3:7.1f} us   clock {g / 8 / d * 1e3:6.0f} MHz   matrix pipes busy {m / (g / 8 * 1024):.3f}")
P
done
