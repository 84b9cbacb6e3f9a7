#!/bin/bash
# Collects the measurement artifacts kept under profiles/ (run on the MI355X box, from the repository root):
#   bash tools/profile_round.sh gpurun_out/rNN && python3 tools/summarize_profiles.py gpurun_out/rNN rNN
# rocprofv3 wants a writable cwd and TMPDIR; PMC passes are separate runs with --kernel-trace only.
set -u
R=$(pwd)
O=$R/${1:-gpurun_out/prof}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 3 --warmup 1 > $O/bench_config1.json 2> $O/bench_config1.err
python3 $R/bench.py --steps 2 --warmup 1 --workload config2 --no-cpu-baseline > $O/bench_config2.json 2> $O/bench_config2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config1_under_rocprof.json 2> $O/kt_c1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2 -- python3 $R/bench.py --steps 1 --warmup 1 --workload config2 --no-cpu-baseline > $O/bench_config2_under_rocprof.json 2> $O/kt_c2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/quick_bench.py 64 12 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/quick_bench.py 64 12 0 > $O/pmc_write.log 2>&1
cut -c1-200 $O/bench_config1.json
cut -c1-200 $O/bench_config2.json
