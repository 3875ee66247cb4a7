#!/bin/bash
# Run ON THE GPU BOX: two-lane 256-tile step time of explicit per-block plans, interleaved, three rounds (KEEP_CALIBRATE=0: no calibration kernels in between).
export KEEP_CALIBRATE=0
for rep in 1 2 3; do
for plan in "attn:224444444444444444444440 mlp:444444444444444444444444" "attn:255444444444444444444440 mlp:444444444444444444444444" "attn:254444444444444444444440 mlp:444444444444444444444444"; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-sustained --no-breakdown --plan "$plan" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$plan', d['value'], d['ms_per_step'])"
done; done
