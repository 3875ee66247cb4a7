#!/bin/bash
# Run ON THE GPU BOX: roofline report of the kernels behind BASELINE config 5 (100 000-tile probability map / similarity): wall time (tools/sim_bench.py), rocprofv3
# kernel stats, FETCH_SIZE / WRITE_SIZE in separate passes (kernel-trace only).   gpurun -- 'tools/config5_roofline.sh' -> gpurun_out/config5_roofline.txt
REPO=$(pwd); OUT=$REPO/gpurun_out/config5_roofline.txt
cd /tmp && export TMPDIR=/tmp
{
echo "BASELINE config 5 (per-tile dense similarity / tumour-probability map over a 100 000-tile slide) -- roofline report of the kernels behind it (library $(sha256sum $REPO/keep_amd/libkeep_hip.so | cut -c1-16))"
echo "tools/sim_bench.py (HIP-sync wall time, 20 calls):"
python $REPO/tools/sim_bench.py 2>/dev/null
echo
echo "rocprofv3 --kernel-trace --stats of the same script (Name, Calls, TotalNs, AverageNs, %, MinNs, MaxNs, StdDev):"
rm -rf /tmp/c5k; rocprofv3 --kernel-trace --stats -d /tmp/c5k --output-format csv -- python $REPO/tools/sim_bench.py > /dev/null 2>&1
grep "sim_" "$(find /tmp/c5k -name '*kernel_stats.csv' | head -1)"
echo
echo "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; KiB per launch; on gfx950 FETCH_SIZE reports half of a wide coalesced read -> x2):"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/c5p; rocprofv3 --kernel-trace --pmc $c -d /tmp/c5p --output-format csv -- python $REPO/tools/sim_bench.py > /dev/null 2>&1
  python - "$(find /tmp/c5p -name '*counter_collection.csv' | head -1)" $c <<'P'
import csv, sys
from collections import defaultdict
a = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "sim_" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]: a[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in a.items(): print(sys.argv[2], k, "mean KiB per launch %.1f launches %d" % (sum(v) / len(v), len(v)))
P
done
} > $OUT 2>&1
cat $OUT
