#!/bin/bash
# Run ON THE GPU BOX: what each ingredient of a 'comp' plan costs in the two-lane 256-tile step (bench.py's timed region, explicit plans, no calibration).
export KEEP_CALIBRATE=0
Z=000000000000000000000000
for rep in 1 2; do
for plan in "attn:$Z mlp:$Z" "attn:220000000000000000000000 mlp:$Z" "attn:$Z mlp:444444444444444444444444" "attn:044444444444444444444440 mlp:444444444444444444444444" "attn:224444444444444444444440 mlp:444444444444444444444444" "attn:244444444444444444444440 mlp:444444444444444444444444"; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-sustained --no-breakdown --plan "$plan" "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$plan', d['value'], d['ms_per_step'])"
done; done
