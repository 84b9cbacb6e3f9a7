"""Turns the raw output of tools/profile_round.sh into the files kept under profiles/.
usage: python3 tools/summarize_profiles.py <raw dir> <tag> [B]      e.g.  gpurun_out/r02 r02 256"""
import collections
import csv
import glob
import os
import shutil
import sys

raw, tag = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
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
busy = gui = None
if glob.glob(os.path.join(raw, "pmc_mfma", "*", "*counter_collection.csv")):
    busy, gui = pmc("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), pmc("pmc_mfma", "GRBM_GUI_ACTIVE")
with open(os.path.join(prof, f"{tag}_hbm_traffic_pmc.csv"), "w") as f:
    f.write("# HBM traffic per launch from rocprofv3 PMC passes (separate passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE)\n")
    f.write(f"# command: rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python3 tools/quick_bench.py {B} 12 0\n")
    f.write("# fetch_MB = 2 * FETCH_SIZE[KB] / 1024 (gfx950 reports half the bytes of 16-B/lane streaming reads, "
            "MI355X_MICROARCH.md HBM);\n")
    f.write("# write_MB = WRITE_SIZE[KB] / 1024.  Median over the launches of the iteration loop; one row = one launch "
            "per iteration.\n")
    f.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs) where collected (third pass)\n")
    f.write("B,kernel,grid_threads,fetch_MB,write_MB,mfma_busy\n")
    for (name, grid), v in fetch.items():
        if len(v) < 12 or "gemm_nt_kernel" in name or "rocclr" in name:      # set-up kernels, autotune runs
            continue
        v, w = sorted(v), sorted(write[(name, grid)])
        mb = ""
        if busy and (name, grid) in busy and (name, grid) in gui:
            bb, gg = sorted(busy[(name, grid)]), sorted(gui[(name, grid)])
            g = gg[len(gg) // 2]
            if g > 0:
                mb = round(bb[len(bb) // 2] / (g / 8.0 * 256 * 4), 3)
        f.write(f'{B},"{name[:100]}",{int(grid)},{round(2 * v[len(v) // 2] / 1024, 1)},{round(w[len(w) // 2] / 1024, 1)},{mb}\n')
src = glob.glob(os.path.join(raw, "kt_c3", "*", "*kernel_stats.csv"))[0]
shutil.copy(src, os.path.join(prof, f"{tag}_bench_config3_kernel_stats.csv"))
for n in ("bench_config3.json", "bench_config3_under_rocprof.json", "bench_config5.json", "bench_config2.json",
          "gemm_power_probe.txt"):
    if os.path.exists(os.path.join(raw, n)):
        shutil.copy(os.path.join(raw, n), os.path.join(prof, f"{tag}_{n}"))
c5 = glob.glob(os.path.join(raw, "kt_c5", "*", "*kernel_stats.csv"))
if c5:
    shutil.copy(c5[0], os.path.join(prof, f"{tag}_bench_config5_kernel_stats.csv"))
print("profiles written for", tag)
