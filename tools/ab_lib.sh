#!/bin/bash
# alternate library variants (variants/lib_<name>.so) on one box: graph-replayed iteration time of tools/quick_bench.py
# usage: tools/ab_lib.sh name1 name2 ... ; QB_ARGS overrides "256 96 1"
for r in 1 2 3; do
  for v in "$@"; do
    AWARE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so python tools/quick_bench.py ${QB_ARGS:-256 96 1} 2>/dev/null | grep "ms/iter" | sed "s/^/$v /"
  done
done
for v in "$@"; do AWARE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so python tools/quick_bench.py 256 32 0 f16x2 prof 2>/dev/null | grep "us per" | sed "s/^/$v /"; done
