#!/bin/bash
# Run ON THE GPU BOX: attention_pers_kernel<13> with half of every SIMD's compute waves starting each pair late (experiment builds -DKEEP_ATTN_STAGGER=n, n x 64 cycles).
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "" _stg8 _stg16 _stg32 _stg64; do
  export KEEP_HIP_LIB=$REPO/keep_amd/libkeep_hip$v.so
  rm -rf /tmp/as
  rocprofv3 --kernel-trace --stats -d /tmp/as --output-format csv -- python $REPO/tools/attn_time.py > /tmp/as.log 2>&1
  echo "lib${v:-_product} $(grep attention_pers "$(find /tmp/as -name '*kernel_stats.csv' | head -1)" | cut -d, -f2-4)"
done; done
