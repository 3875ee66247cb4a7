#!/bin/bash
# A/B an engine option on ONE box (boxes differ by a few %): tools/ab_bench.sh name v0 v1 [reps]
name=$1; v0=$2; v1=$3; reps=${4:-3}
for i in $(seq $reps); do
  for v in $v0 $v1; do
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --opt $name=$v 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name=$v', d['value'], d['ms_per_step'])"
  done
done
