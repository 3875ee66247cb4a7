#!/bin/bash
# tools/ab_lib.sh libA.so libB.so ... : bench.py with each experiment build (KEEP_HIP_LIB), interleaved, on one box
reps=3
for i in $(seq $reps); do
  for l in "$@"; do
    KEEP_HIP_LIB=$PWD/keep_amd/$l timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'])"
  done
done
