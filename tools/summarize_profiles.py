"""Turns the raw output of tools/profile_round.sh into the files kept under profiles/.
usage: python3 tools/summarize_profiles.py <raw dir> <tag> [B]      e.g.  gpurun_out/r03 r03 256"""
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
sys.path.insert(0, root)
from bench import kernel_source_hash  # noqa: E402


def pmc(dirname, counter):
    paths = glob.glob(os.path.join(raw, dirname, "*", "*counter_collection.csv"))
    agg = collections.OrderedDict()
    if not paths:
        return agg
    for r in csv.DictReader(open(paths[0])):
        if r["Counter_Name"] == counter:
            agg.setdefault((r["Kernel_Name"], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    return agg


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
busy, gui = pmc("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), pmc("pmc_mfma", "GRBM_GUI_ACTIVE")
with open(os.path.join(prof, f"{tag}_hbm_traffic_pmc.csv"), "w") as f:
    f.write(f"# kernel_source_hash={kernel_source_hash()}  (sha256 over aware_amd/csrc/*.hip|hpp|h; bench.py quotes this table only on the same sources)\n")
    f.write("# HBM traffic per launch from rocprofv3 PMC passes (separate passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE)\n")
    f.write(f"# command: rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python3 tools/quick_bench.py {B} 12 0\n")
    f.write("# fetch_MB = 2 * FETCH_SIZE[KB] / 1024 (gfx950 reports half the bytes of 16-B/lane streaming reads, "
            "MI355X_MICROARCH.md HBM);\n")
    f.write("# write_MB = WRITE_SIZE[KB] / 1024.  Median over the launches of the iteration loop; one row = one launch "
            "per iteration.\n")
    f.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs) where collected (third pass)\n")
    f.write("B,kernel,grid_threads,fetch_MB,write_MB,mfma_busy\n")
    for (name, grid), v in fetch.items():
        if len(v) < 12 or "gemm_nt_kernel" in name or "rocclr" in name or (name, grid) not in write:      # set-up kernels
            continue
        mb = ""
        if (name, grid) in busy and (name, grid) in gui and med(gui[(name, grid)]) > 0:
            mb = round(med(busy[(name, grid)]) / (med(gui[(name, grid)]) / 8.0 * 256 * 4), 3)
        f.write(f'{B},"{name[:100]}",{int(grid)},{round(2 * med(v) / 1024, 1)},{round(med(write[(name, grid)]) / 1024, 1)},{mb}\n')

# issue-side counters per kernel of the loop (medians; SQ_* cycle counters are quad-cycles summed over the waves)
sq = {}
for d, names in (("pmc_sq1", ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                              "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"]),
                 ("pmc_sq2", ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU",
                              "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS"]),
                 ("pmc_mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])):
    for c in names:
        for key, v in pmc(d, c).items():
            if len(v) >= 12:
                sq.setdefault(key, {})[c] = med(v)
if sq:
    cols = ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
            "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS",
            "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"]
    with open(os.path.join(prof, f"{tag}_issue_counters_pmc.csv"), "w") as f:
        f.write(f"# kernel_source_hash={kernel_source_hash()}\n")
        f.write(f"# medians per launch over the iteration loop of tools/quick_bench.py {B} 12 0 (three rocprofv3 --pmc passes);\n")
        f.write("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over\n")
        f.write("# SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs (MI355X_MICROARCH.md, per-instruction constants / DVFS give-back)\n")
        f.write("kernel,grid_threads," + ",".join(cols) + "\n")
        for (name, grid), d in sq.items():
            if "gemm_nt_kernel" in name or "rocclr" in name:
                continue
            f.write(f'"{name[:100]}",{int(grid)},' + ",".join(str(int(d.get(c, -1))) for c in cols) + "\n")

src = glob.glob(os.path.join(raw, "kt_c3", "*", "*kernel_stats.csv"))[0]
shutil.copy(src, os.path.join(prof, f"{tag}_bench_config3_kernel_stats.csv"))
for n in ("bench_config3.json", "bench_config3_under_rocprof.json", "bench_config3_l1.json", "bench_config5.json", "bench_config2.json",
          "drift.json"):
    if os.path.exists(os.path.join(raw, n)):
        shutil.copy(os.path.join(raw, n), os.path.join(prof, f"{tag}_{n}"))
c5 = glob.glob(os.path.join(raw, "kt_c5", "*", "*kernel_stats.csv"))
if c5:
    shutil.copy(c5[0], os.path.join(prof, f"{tag}_bench_config5_kernel_stats.csv"))

# K-loop ablations of the conv GEMM: average duration of the five launches per variant
abl = sorted(glob.glob(os.path.join(raw, "abl_*", "*", "*kernel_stats.csv")))
if abl:
    with open(os.path.join(prof, f"{tag}_gemm_h2_ablation.txt"), "w") as f:
        f.write("Timing-only ablations of gemm_clip_h2_kernel's K loop (results invalid; tools/build_variant.sh ... -DH2_ABL=bits,\n"
                "tools/h2_time.py 256 94 under rocprofv3 --kernel-trace --stats): average microseconds per launch.\n"
                "bits: 1 no split arithmetic, 2 no LDS stores, 4 no fragment reads, 8 no weight-fragment loads, 16 no barrier,\n"
                "32 no global A loads; 63 = MFMAs + epilogue only.\n"
                "Read the rows as UPPER bounds of what removing a part would save: a variant without the split feeds raw f32 bit\n"
                "patterns (NaNs among them) to the matrix pipe, one without loads keeps multiplying the same registers -- both draw\n"
                "less power than live data and may run at a higher DVFS clock.  profiles/r03_gemm_planes_experiment.txt has the\n"
                "same ablations on finite, changing operands (MFMAs only: 1.18 PFLOP/s issued, the chip's sustained rate).\n\n")
        for path in abl:
            v = path.split(os.sep)[-3][len("abl_"):]
            rows = [r for r in csv.DictReader(open(path)) if "gemm_clip_h2_kernel" in r["Name"]]
            f.write(f"{v:24s} " + "  ".join(f'{r["Name"].split("<")[1].split(">")[0]}: {float(r["AverageNs"]) / 1e3:7.1f} us x{r["Calls"]}' for r in rows) + "\n")
# clock the chip holds inside each kernel of the loop: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration
mf = glob.glob(os.path.join(raw, "pmc_mfma", "*", "*counter_collection.csv"))
if mf:
    import collections
    import statistics
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(mf[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "aware::" not in r["Kernel_Name"]:
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if dur > 0:
            acc[(r["Kernel_Name"], r["Grid_Size"])].append((float(r["Counter_Value"]) / 8.0 / dur, dur))
    with open(os.path.join(prof, f"{tag}_kernel_clocks.csv"), "w") as f:
        f.write("# effective clock per kernel of the embed loop = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (rocprofv3 --pmc\n"
                "# SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace, tools/quick_bench.py " + str(B) + " 12 0; medians over the loop's launches).\n"
                "# MI355X_MICROARCH.md, DVFS give-back: the quotient reads high on dispatches shorter than about 0.3 ms -- compare kernels of\n"
                "# similar length: the conv GEMMs hold 1.8-1.9 GHz, the HBM-bound DSP kernels of the same length 2.4-2.5 GHz.\n"
                "kernel,grid_threads,median_us,effective_GHz,launches\n")
        rows_ = [(statistics.median(x[1] for x in v), k, statistics.median(x[0] for x in v), len(v)) for k, v in acc.items() if len(v) >= 8]
        for dur, k, clk, n in sorted(rows_, reverse=True):
            f.write(f'"{k[0][:100]}",{k[1]},{dur / 1e3:.1f},{clk:.2f},{n}\n')
print("profiles written for", tag)
