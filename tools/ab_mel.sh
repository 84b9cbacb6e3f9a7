#!/bin/bash
# alternate the two forms of the mel backward on one box: graph-replayed iteration time, then the per-kind breakdown
for r in 1 2 3; do
  for m in dense taps; do
    QB_MEL=$m python tools/quick_bench.py 256 96 1 2>/dev/null | grep "ms/iter" | sed "s/^/$m /"
  done
done
for m in dense taps; do QB_MEL=$m python tools/quick_bench.py 256 32 0 f16x2 prof 2>/dev/null | grep "us per" | sed "s/^/$m /"; done
