#!/bin/bash
# tools/ab_run.sh OUTDIR lib1.so lib2.so ... : interleaved A/B of experiment builds on one box (tools/ab_features.py: 20 steps of 256 tiles,
# setting 1/8, no calibration), three rounds; the features of round 1 are kept for bit-comparison against the first library
out=$1; shift
mkdir -p $out
for i in 1 2 3; do
  for l in "$@"; do
    n=${l%.so}
    if [ $i = 1 ]; then KEEP_HIP_LIB=$PWD/keep_amd/$l timeout 300 python tools/ab_features.py --out $out/$n.pt 2>/dev/null
    else KEEP_HIP_LIB=$PWD/keep_amd/$l timeout 300 python tools/ab_features.py 2>/dev/null; fi
  done
done
first=${1%.so}
for l in "$@"; do n=${l%.so}; echo -n "$n vs $first: "; python tools/ab_features.py --compare $out/$first.pt $out/$n.pt; done
