#!/bin/bash
# tiles/s against batch size and number of internal stream lanes, on one box
for s in 1 2 3 4; do
  for b in 64 128 192 256 384 512 768; do
    timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-breakdown --batch $b --opt max_tiles=$b --opt streams=$s 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$s batch=$b', d['value'], d['ms_per_step'])"
  done
done
