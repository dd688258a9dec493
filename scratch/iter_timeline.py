"""Timeline of one graph-replayed iteration from a rocprofv3 kernel trace csv.
usage: python scratch/iter_timeline.py <ks_kernel_trace.csv> [iteration index, default 20] [min_us]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
its, cur = [], None
for r in rows:
    if "dropout_mask_multi" in r["Kernel_Name"] or (cur is None and "attr_pack" in r["Kernel_Name"]):
        cur = []
        its.append(cur)
    if cur is not None:
        cur.append(r)
it = its[min(k, len(its) - 1)]
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("ali::", "")[:58]
t0 = int(it[0]["Start_Timestamp"])
fam = {}
for i, r in enumerate(it):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = short(r["Kernel_Name"])
    fam.setdefault(n, [0, 0.0])
    fam[n][0] += 1
    fam[n][1] += d
    print(f"{i:3d} {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d:7.1f} {n} grid {r['Grid_Size_X']}")
print(f"--- {len(it)} launches, {(int(it[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
for n, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:58s} {c:4d} {t:8.1f} us")
