"""Per-kernel average duration from rocprofv3 --kernel-trace result databases.  usage: trace_summary.py DIR_OR_DB [filter]"""
import sys, glob, os, sqlite3, re
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by name order by 5 desc").fetchall()
    print("==", db)
    for name, n, avg, mn, tot in rows:
        if flt and flt not in name:
            continue
        short = re.sub(r"\(.*", "", name)
        print(f"{n:6d} avg {avg / 1e3:9.1f} us  min {mn / 1e3:9.1f}  total {tot / 1e6:9.2f} ms  {short[:100]}")
