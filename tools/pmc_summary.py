"""Per-kernel mean counter values from rocprofv3 --pmc result databases.  usage: pmc_summary.py DIR [kernel substring]"""
import sys, glob, os, sqlite3, re, collections
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(db).cursor()
    for name, grid, wg, cname, val in cur.execute("select kernel_name, grid_size_x, workgroup_size_x, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name"):
        if flt and flt not in name:
            continue
        key = (re.sub(r"\(.*", "", name)[:70], grid // max(wg, 1), wg)
        agg[key][cname].append(val)
for key in sorted(agg):
    print("==", key)
    for c in sorted(agg[key]):
        v = agg[key][c]
        print(f"   {c:34s} {sum(v) / len(v):16.0f}   (n={len(v)})")
