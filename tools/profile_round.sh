#!/bin/bash
# Collects the measurement artifacts kept under profiles/ (run on the MI355X box, from the repository root):
#   bash tools/profile_round.sh gpurun_out/rNN && python3 tools/summarize_profiles.py gpurun_out/rNN rNN
# rocprofv3 wants a writable cwd and TMPDIR; PMC passes are separate runs with --kernel-trace only; the program after `--`
# is python3 itself (no env / shell wrapper).  B = clips per GPU of the PMC passes (256 = the default bench workload).
set -u
R=$(pwd)
O=$R/${1:-gpurun_out/prof}
B=${2:-256}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
echo "[profile] bench lines"
python3 $R/bench.py --steps 3 --warmup 1 > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
python3 $R/bench.py --workload config3_l1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config3_l1.json 2> $O/bench_config3_l1.err || exit 1
python3 $R/bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err || exit 1
python3 $R/bench.py --workload config2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config2.json 2> $O/bench_config2.err || exit 1
echo "[profile] kernel traces"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config3_under_rocprof.json 2> $O/kt_c3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -- python3 $R/bench.py --workload config5 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/bench_config5_under_rocprof.json 2> $O/kt_c5.err || exit 1
echo "[profile] counter passes"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/quick_bench.py $B 12 0 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/quick_bench.py $B 12 0 > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/quick_bench.py $B 12 0 > $O/pmc_mfma.log 2>&1 || exit 1
# issue-side counters of the loop's kernels (the conv GEMM's budget)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 $R/tools/quick_bench.py $B 12 0 > $O/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 $R/tools/quick_bench.py $B 12 0 > $O/pmc_sq2.log 2>&1 || exit 1
# timing-only ablations of the conv GEMM's K loop, when the variant libraries were built (tools/build_variant.sh, -DH2_ABL=...)
if ls $R/variants/lib_h2abl_*.so > /dev/null 2>&1; then
  echo "[profile] K-loop ablations"
  for lib in $R/variants/lib_h2abl_*.so; do
    v=$(basename $lib .so); v=${v#lib_}
    export AWARE_HIP_LIB=$lib
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$v -- python3 $R/tools/h2_time.py $B 94 2 > $O/abl_$v.log 2>&1 || exit 1
    unset AWARE_HIP_LIB
  done
fi
python3 $R/tests/tools/measure_drift.py > $O/drift.log 2>&1 && cp $R/gpurun_out/drift.json $O/drift.json
cut -c1-300 $O/bench_config3.json
