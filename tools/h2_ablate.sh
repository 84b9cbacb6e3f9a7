#!/bin/bash
# run tools/h2_time.py under the kernel trace for every variants/lib_<name>.so named on the command line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abl
for v in "$@"; do
  export AWARE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so
  timeout -k 10 120 rocprofv3 --kernel-trace -d gpurun_out/abl/$v -o t -- python3 tools/h2_time.py 256 94 2 > gpurun_out/abl/$v.log 2>&1 || exit 1
  echo "$v done"
done
