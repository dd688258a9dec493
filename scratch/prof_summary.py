"""Aggregate a rocprofv3 rocpd sqlite result (kernel trace) into per-kernel totals per iteration."""
import sqlite3, sys, re, glob
path = sys.argv[1]; iters = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
db = sqlite3.connect(path); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.display_name, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.display_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total {tot/1e6/iters:.3f} ms/iter, launches/iter {sum(r[1] for r in rows)/iters:.1f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    n = re.sub(r"at::native::|\(anonymous namespace\)::", "", r[0])
    print(f"{n[:100]:100s} {r[1]/iters:6.1f}/it {r[2]/1e3/iters:8.1f}us/it {100*r[2]/tot:5.1f}% avg {r[2]/r[1]/1e3:7.1f}us")
