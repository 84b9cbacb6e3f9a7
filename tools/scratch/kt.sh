cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ro in 1 0; do
export AWARE_TUNE_READOUT=$ro
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$ro -- python3 $R/tools/quick_bench.py 64 32 0 > $R/gpurun_out/kt_$ro.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/kt_$ro/*/*kernel_stats.csv")[0]
print("readout fused =", $ro)
for r in list(csv.DictReader(open(f))):
    if "gemm_nt_kernel" in r["Name"] and int(r["Calls"]) < 60: continue
    if float(r["Percentage"]) < 0.5: continue
    print("  ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
