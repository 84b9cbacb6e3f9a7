#!/bin/bash
# A/B of library variants on the embed loop in ONE box visit, alternating: tools/ab_loop.sh ROUNDS name1 name2 ...
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    export AWARE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so
    echo -n "$v: "; timeout -k 10 120 python tools/quick_bench.py 256 128 1 f16x2 2>&1 | grep "ms/iter"
  done
done
