#!/bin/bash
# Run ON THE GPU BOX (via gpurun): regenerates the evidence files that profiles/ keeps.
#   gpurun -- 'tools/refresh_profiles.sh'   ->   gpurun_out/prof/*   (copy what is to be kept to profiles/rNN_*)
# Every profiled pass runs the same command as the plain bench with the calibration switched off and the settings the plain run's calibration
# chose given explicitly, so a trace holds encode steps only (not the calibration's strict / ladder passes).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof; rm -rf "$OUT"; mkdir -p "$OUT"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"
PLAN=$(python -c "import json;print(json.load(open('$OUT/bench.json'))['config']['comp_settings']['plan'])")
SET=(--plan "$PLAN")
LEAN="--no-cpu-baseline --no-configs --no-sustained"
export KEEP_CALIBRATE=0
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel time summary of the bench command (two internal lanes, as the bench runs)
rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python "$REPO/bench.py" --steps 10 --warmup 3 $LEAN "${SET[@]}" \
    > "$OUT/bench_under_rocprofv3.json" 2> "$OUT/kt.log"
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
# 2. the same on ONE internal stream: per-kernel durations with nothing else on the GPU (what bench.py's roofline.single_stream figures are made of)
rocprofv3 --kernel-trace --stats -d /tmp/kt1 --output-format csv -- python "$REPO/bench.py" --steps 10 --warmup 3 $LEAN --no-breakdown --opt streams=1 "${SET[@]}" \
    > "$OUT/bench_single_stream_under_rocprofv3.json" 2> "$OUT/kt1.log"
cp "$(find /tmp/kt1 -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_single_stream.csv"
# 3. counters: their own passes, kernel-trace only (FETCH_SIZE and WRITE_SIZE do not share a pass); one internal stream, so a launch is 256 tiles --
#    the same launch bench.py's roofline.flops_per_launch / avg_launch_ms describe
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c --output-format csv -- python "$REPO/bench.py" --steps 3 --warmup 1 $LEAN --no-breakdown --opt streams=1 "${SET[@]}" \
      > /dev/null 2> "$OUT/pmc_$c.log"
done
# 4. matrix-pipe utilisation and clock, two lanes and one stream (clean per-kernel attribution) -> the per-kernel table
for v in "" "1"; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_MFMA$v --output-format csv -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 $LEAN --no-breakdown "${SET[@]}" ${v:+--opt streams=1} > /dev/null 2> "$OUT/pmc_MFMA$v.log"
done
cd "$REPO"
python tools/pmc_summary.py "$(find /tmp/pmc_MFMA -name '*counter_collection.csv' | head -1)" > "$OUT/mfma_busy.txt" 2>&1
sha256sum keep_amd/libkeep_hip.so | cut -c1-16 > "$OUT/lib_sha16.txt"
python tools/pmc_traffic.py "$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
                            "$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$OUT/hbm_traffic.json" 256 > /dev/null
head -1 "$(find /tmp/pmc_MFMA1 -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_csv_header.txt"
# the raw single-stream counter CSV travels back too (a few MB), so that tools/kernel_table.py can be re-run off the box
gzip -c "$(find /tmp/pmc_MFMA1 -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_mfma_single_stream_counter_collection.csv.gz"
# encode calls in the single-stream trace: 3 warm-up + 10 timed + 3 clock-probe steps
python tools/kernel_table.py "$OUT/kernel_stats_single_stream.csv" "$(find /tmp/pmc_MFMA1 -name '*counter_collection.csv' | head -1)" "$OUT/hbm_traffic.json" 16 "$OUT/bench.json" \
    > "$OUT/per_kernel_table.md" 2> "$OUT/per_kernel_table.err"
# 5. the text tower (SURVEY.md 8d asks prompts/s beside tiles/s): config 3's 64 x 256 prompt bank, at the trimmed and at the padded length
cd /tmp
for t in 1 0; do
  rocprofv3 --kernel-trace --stats -d /tmp/kt_text$t --output-format csv -- python "$REPO/tools/text_profile.py" --trim $t > "$OUT/text_trim$t.txt" 2> "$OUT/kt_text$t.log"
  cp "$(find /tmp/kt_text$t -name '*kernel_stats.csv' | head -1)" "$OUT/text_kernel_stats_trim$t.csv"
done
cd "$REPO"
unset KEEP_CALIBRATE
python tools/clock_check.py > "$OUT/clock_under_load.txt" 2>&1
cat "$OUT/per_kernel_table.md"; cat "$OUT/pmc_csv_header.txt"
head -c 400 "$OUT/bench.json"; echo; head -4 "$OUT/kernel_stats_single_stream.csv"
