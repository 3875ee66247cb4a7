#!/bin/bash
# tools/sweep_opt.sh name v1 v2 ... : bench.py once per value of one engine option, on one box
name=$1; shift
for v in "$@"; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --opt $name=$v 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name=$v', d['value'], d['ms_per_step'])"
done
