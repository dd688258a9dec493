"""top kernels of a rocprofv3 kernel_stats.csv: python scratch/kshare.py <csv> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 26
tot = sum(float(r["TotalDurationNs"]) for r in rows)
acc = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    nm = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("ali::", "")[:58]
    t = float(r["TotalDurationNs"]); acc += t; c = int(r["Calls"])
    print("%-58s calls %5d avg %9.1f us %5.1f%% cum %5.1f%%" % (nm, c, t / c / 1e3, 100 * t / tot, 100 * acc / tot))
