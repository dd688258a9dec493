"""Per-iteration kernel table from a rocprofv3 --kernel-trace run (results.db or *_kernel_stats.csv).
usage: python scratch/kstats.py <db-or-csv> <iterations> [top]"""
import csv, re, sqlite3, sys
path, iters = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = []
if path.endswith(".db"):
    cur = sqlite3.connect(path).cursor()
    rows = [(n, c, t) for n, c, t in cur.execute("select name, count(*), sum(end-start) from kernels group by name")]
else:
    for r in csv.DictReader(open(path)):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])))
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "").replace("ali::", "")
    return n[-64:]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print(f"total {tot/1e6/iters:.3f} ms/iter, {sum(r[1] for r in rows)/iters:.0f} launches/iter")
for n, c, t in rows[:top]:
    print(f"{short(n):64s} {c/iters:7.1f}/it {t/1e3/iters:9.1f} us/it {t/c/1e3:8.1f} us avg")
