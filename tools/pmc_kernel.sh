#!/bin/bash
# Run ON THE GPU BOX: mean value of a list of hardware counters per launch of the kernels matching a pattern (one rocprofv3 --pmc pass per group of counters).
#   gpurun -- 'tools/pmc_kernel.sh attention_pers "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"'
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_kernel; mkdir -p "$OUT"
PAT=$1; shift
export KEEP_CALIBRATE=0
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  rm -rf /tmp/pmck_$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmck_$i --output-format csv -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-sustained --no-breakdown --opt streams=1 --precision fp16 > /dev/null 2> "$OUT/grp$i.log"
  python - "$(find /tmp/pmck_$i -name '*counter_collection.csv' | head -1)" "$PAT" <<'P' | tee -a "$OUT/summary.txt"
import csv, sys
from collections import defaultdict
agg = defaultdict(list); dur = []
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r: dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for c, v in sorted(agg.items()):
    print(f"{sys.argv[2]:24s} {c:28s} mean/launch {sum(v) / len(v):16.1f}   launches {len(v)}   (duration {sum(dur) / max(len(dur), 1) / 1e3:.1f} us)")
P
  i=$((i+1))
done
