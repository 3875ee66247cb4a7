#!/bin/bash
# tools/ab_quick.sh "<label>|<env assignments>|<bench args>" ... : the timed region of bench.py only (20 steps), each arm 3 times, interleaved, one box
reps=${REPS:-3}
for i in $(seq $reps); do
  for arm in "$@"; do
    IFS='|' read -r label envs args <<< "$arm"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --no-configs --no-sustained $args 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['config']['comp_settings']['comp_full_blocks'], d['config']['comp_settings']['comp_mlp_blocks'])"
  done
done
