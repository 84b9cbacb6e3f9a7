#!/bin/bash
# Collects the measurement artifacts kept under profiles/ (run on the MI355X box, from the repository root):
#   bash tools/profile_round.sh gpurun_out/rNN && python3 tools/summarize_profiles.py gpurun_out/rNN rNN
# rocprofv3 wants a writable cwd and TMPDIR; PMC passes are separate runs with --kernel-trace only.
# B = clips per GPU of the PMC passes (256 = the default bench workload, BASELINE config 3).
set -u
R=$(pwd)
O=$R/${1:-gpurun_out/prof}
B=${2:-256}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 3 --warmup 1 > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config3_under_rocprof.json 2> $O/kt_c3.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/quick_bench.py $B 12 0 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/quick_bench.py $B 12 0 > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/quick_bench.py $B 12 0 > $O/pmc_mfma.log 2>&1 || exit 1
# BASELINE config 5 (ragged 1-10 s clips, per-clip chains): the line and the kernel table
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -- python3 $R/bench.py --workload config5 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/bench_config5_under_rocprof.json 2> $O/kt_c5.err || exit 1
python3 $R/bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $O/bench_config5.json 2> $O/bench_config5.err || exit 1
python3 $R/bench.py --workload config2 --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/bench_config2.json 2> $O/bench_config2.err || exit 1
python3 $R/tools/gemm_power_probe.py 1024 40 > $O/gemm_power_probe.txt 2>&1 || exit 1
cut -c1-300 $O/bench_config3.json
