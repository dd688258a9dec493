"""Aggregate a rocprofv3 --pmc csv (…_counter_collection.csv) per kernel: launches and mean counter values.
usage: python scratch/pmc_summary.py <counter_collection.csv> [min_launches]"""
import csv, re, sys
from collections import defaultdict
path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(path)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ali::", "")
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[n][r["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
print("kernel,launches," + ",".join(names))
for k in sorted(acc, key=lambda k: -max(acc[k].values())):
    n = max(cnt[k].values())
    if len(sys.argv) > 2 and n < int(sys.argv[2]):
        continue
    print("\"" + k[-60:] + "\"," + str(n) + "," + ",".join(f"{acc[k][c] / max(cnt[k][c], 1):.4g}" for c in names))
