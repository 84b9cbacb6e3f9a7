"""Turns the raw output of tools/profile_round.sh into the files kept under profiles/.
usage: python3 tools/summarize_profiles.py <raw dir> <tag>      e.g.  gpurun_out/r01 r01"""
import collections
import csv
import glob
import os
import shutil
import sys

raw, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")


def pmc(dirname, counter):
    path = glob.glob(os.path.join(raw, dirname, "*", "*counter_collection.csv"))[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg.setdefault((r["Kernel_Name"], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    return agg


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
with open(os.path.join(prof, f"{tag}_hbm_traffic_pmc.csv"), "w") as f:
    f.write("# HBM traffic per launch from rocprofv3 PMC passes (separate passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE)\n")
    f.write("# command: rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python3 tools/quick_bench.py 64 12 0\n")
    f.write("# fetch_MB = 2 * FETCH_SIZE[KB] / 1024 (gfx950 reports half the bytes of 16-B/lane streaming reads, "
            "MI355X_MICROARCH.md HBM);\n")
    f.write("# write_MB = WRITE_SIZE[KB] / 1024.  Median over the launches of the iteration loop; one row = one launch "
            "per iteration.\n")
    f.write("B,kernel,grid_threads,fetch_MB,write_MB\n")
    for (name, grid), v in fetch.items():
        if len(v) < 12 or "gemm_nt_kernel" in name or "rocclr" in name:      # set-up kernels, autotune runs
            continue
        v, w = sorted(v), sorted(write[(name, grid)])
        f.write(f'64,"{name[:100]}",{int(grid)},{round(2 * v[len(v) // 2] / 1024, 1)},{round(w[len(w) // 2] / 1024, 1)}\n')
for cfg in ("c1", "c2"):
    src = glob.glob(os.path.join(raw, f"kt_{cfg}", "*", "*kernel_stats.csv"))[0]
    shutil.copy(src, os.path.join(prof, f"{tag}_bench_config{cfg[1]}_kernel_stats.csv"))
for n in ("bench_config1.json", "bench_config2.json", "bench_config1_under_rocprof.json"):
    shutil.copy(os.path.join(raw, n), os.path.join(prof, f"{tag}_{n}"))
print("profiles written for", tag)
