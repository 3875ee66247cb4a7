#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of the small-M kernels (CLS-row chain) with the 128x128 kernel on (skinny_wide=1) and off, one internal stream, explicit plan.
set -u
REPO=$(pwd); mkdir -p gpurun_out/skinny
export KEEP_CALIBRATE=0
cd /tmp && export TMPDIR=/tmp
for w in 1 0; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks$w --output-format csv -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-sustained --no-breakdown --opt streams=1 --opt skinny_wide=$w --plan "attn:224444444444444444444440 mlp:444444444444444444444444" > /dev/null 2> /tmp/ks$w.log
  cp "$(find /tmp/ks$w -name '*kernel_stats.csv' | head -1)" "$REPO/gpurun_out/skinny/kernel_stats_wide$w.csv"
  echo "== skinny_wide=$w"; grep -i "skinny\|gather\|scatter\|layernorm_kernel" "$REPO/gpurun_out/skinny/kernel_stats_wide$w.csv" | cut -c1-200
done
