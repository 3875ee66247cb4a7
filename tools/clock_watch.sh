#!/bin/bash
# shader clock / power while bench.py runs (run on the GPU box): tools/clock_watch.sh [bench args]
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-breakdown "$@" > /tmp/cw_bench.json 2>/dev/null &
pid=$!
sleep 8
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | head -4
  sleep 0.7
done
wait $pid
python -c "import json; d=json.loads(open('/tmp/cw_bench.json').read()); print('bench', d['value'], d['ms_per_step'])"
