#!/bin/bash
# Run ON THE GPU BOX (via gpurun): regenerates the evidence files that profiles/ keeps.
#   gpurun -- 'tools/refresh_profiles.sh'   ->   gpurun_out/prof/{bench.json,kernel_stats.csv,bench_under_rocprofv3.json,hbm_traffic.json}
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof; rm -rf "$OUT"; mkdir -p "$OUT"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-sustained \
    > "$OUT/bench_under_rocprofv3.json" 2> "$OUT/kt.log"
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
# the same command on ONE internal stream: per-kernel durations with nothing else on the GPU (what bench.py's `single_stream` figures are made of)
rocprofv3 --kernel-trace --stats -d /tmp/kt1 --output-format csv -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-sustained --no-breakdown --opt streams=1 \
    > "$OUT/bench_single_stream_under_rocprofv3.json" 2> "$OUT/kt1.log"
cp "$(find /tmp/kt1 -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_single_stream.csv"
# counters: their own passes, kernel-trace only (FETCH_SIZE and WRITE_SIZE do not share a pass)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c --output-format csv -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-configs --no-sustained \
      > /dev/null 2> "$OUT/pmc_$c.log"
done
# MFMA pipe utilisation of the same command (its own pass)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_MFMA --output-format csv -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-configs --no-sustained \
    > /dev/null 2> "$OUT/pmc_MFMA.log"
cd "$REPO"
python tools/pmc_summary.py "$(find /tmp/pmc_MFMA -name '*counter_collection.csv' | head -1)" > "$OUT/mfma_busy.txt" 2>&1
# the same counters on ONE internal stream (clean per-kernel attribution) -> the per-kernel table
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_MFMA1 --output-format csv -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-configs --no-sustained --opt streams=1 \
    > /dev/null 2> "$OUT/pmc_MFMA1.log"
cd "$REPO"
sha256sum keep_amd/libkeep_hip.so | cut -c1-16 > "$OUT/lib_sha16.txt"
python tools/pmc_traffic.py "$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
                            "$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$OUT/hbm_traffic.json" > /dev/null
python tools/kernel_table.py "$OUT/kernel_stats_single_stream.csv" "$(find /tmp/pmc_MFMA1 -name '*counter_collection.csv' | head -1)" "$OUT/hbm_traffic.json" 13 > "$OUT/per_kernel_table.md" 2> "$OUT/per_kernel_table.err"
python tools/clock_check.py > "$OUT/clock_under_load.txt" 2>&1
cat "$OUT/per_kernel_table.md"
head -c 600 "$OUT/bench.json"; echo; head -5 "$OUT/kernel_stats.csv"; cat "$OUT/hbm_traffic.json" | head -12
