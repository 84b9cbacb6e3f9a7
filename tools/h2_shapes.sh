#!/bin/bash
# correctness (fp64 / oracle tests) + kernel-trace timing of every h2 wave-arrangement variant named on the command line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abl
for v in "$@"; do
  export AWARE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so
  timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm_clip or large_uniform or golden_trajectory_inside" > gpurun_out/abl/$v.test.log 2>&1; echo "$v tests rc=$?"; tail -2 gpurun_out/abl/$v.test.log
  rm -rf gpurun_out/abl/$v
  timeout -k 10 120 rocprofv3 --kernel-trace -d gpurun_out/abl/$v -o t -- python3 tools/h2_time.py 256 94 2 > gpurun_out/abl/$v.log 2>&1 || exit 1
  echo "$v timed"
done
