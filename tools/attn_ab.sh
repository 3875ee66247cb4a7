#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of tools/attn_time.py for the experiment builds keep_amd/libkeep_hip_trim{0,1,2}.so (-DKEEP_ATTN_TRIM=v, tools/experiments/attention_softmax_trim.patch applied).
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1 2; do
  export KEEP_HIP_LIB=$REPO/keep_amd/libkeep_hip_trim$v.so
  rm -rf /tmp/at$v
  rocprofv3 --kernel-trace --stats -d /tmp/at$v --output-format csv -- python $REPO/tools/attn_time.py > /tmp/at$v.log 2>&1
  echo "trim=$v $(grep attention_pers "$(find /tmp/at$v -name '*kernel_stats.csv' | head -1)" | cut -d, -f2-5) $(grep checksum /tmp/at$v.log | sed 's/.*checksum//' | tr '\n' ' ')"
done; done
