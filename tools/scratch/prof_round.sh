set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01b
mkdir -p $O
python3 $R/bench.py --steps 3 --warmup 1 > $O/bench_config1.json 2> $O/bench_config1.err
python3 $R/bench.py --steps 2 --warmup 1 --workload config2 --no-cpu-baseline > $O/bench_config2.json 2> $O/bench_config2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config1_under_rocprof.json 2> $O/kt_c1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/quick_bench.py 64 12 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/quick_bench.py 64 12 0 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/quick_bench.py 64 12 0 > $O/pmc_mfma.log 2>&1
ls $O
cut -c1-300 $O/bench_config1.json
cut -c1-300 $O/bench_config2.json
