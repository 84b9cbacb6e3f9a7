#!/bin/bash
# SQ counter passes over tools/h2_time.py for one library variant (default: the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-main}
[ "$V" != main ] && export AWARE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$V.so
mkdir -p gpurun_out/pmc_$V
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d gpurun_out/pmc_$V/p$i -o c -- python3 tools/h2_time.py 256 94 1 > gpurun_out/pmc_$V/p$i.log 2>&1 || { tail -5 gpurun_out/pmc_$V/p$i.log; }
  echo "pass $i done"
done
