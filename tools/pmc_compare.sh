#!/bin/bash
# Run ON THE GPU BOX: matrix-pipe busy share and effective clock of the GEMM kernels under two option sets (one internal stream, plain-fp16 plan).
#   gpurun -- 'tools/pmc_compare.sh "name=value ..." "name=value ..."'  ->  gpurun_out/pmc_compare/*.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_compare; mkdir -p "$OUT"
export KEEP_CALIBRATE=0
cd /tmp && export TMPDIR=/tmp
i=0
for arm in "$@"; do
  opts=(); for kv in $arm; do opts+=(--opt "$kv"); done
  rm -rf /tmp/pmcc_$i
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcc_$i --output-format csv -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-sustained --no-breakdown --opt streams=1 --precision fp16 "${opts[@]}" > /dev/null 2> "$OUT/arm$i.log"
  python - "$(find /tmp/pmcc_$i -name '*counter_collection.csv' | head -1)" "$arm" > "$OUT/arm$i.txt" <<'P'
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "gemm_f16_v2" not in k: continue
    agg[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        agg[k.split("(")[0][-60:]]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("arm:", sys.argv[2])
for k, c in sorted(agg.items()):
    g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]); m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    d = sum(c["dur_ns"]) / len(c["dur_ns"]) if c["dur_ns"] else float("nan")
    print(f"{k:62s} n {len(c['GRBM_GUI_ACTIVE']):4d}  dur {d / 1e3:8.1f} us  clock {g / 8 / d * 1e3:7.0f} MHz  mfma busy {m / (g / 8 * 1024):.3f}")
P
  cat "$OUT/arm$i.txt"
  i=$((i+1))
done
