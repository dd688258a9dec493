"""gpurun_out/r03prof_<workload>_<precision>/ (scratch/r03_collect.sh) -> the small tracked summaries under profiles/:
  r03_bench_<w>_<p>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of `python3 bench.py --workload w --precision p`
  r03_bench_<w>_<p>_under_rocprof.json    the JSON line that run printed
  r03_pmc_hbm_traffic_<w>_<p>.csv         FETCH_SIZE / WRITE_SIZE per kernel and launch (separate --pmc passes), what bench.py reads
  r03_pmc_tcc_<w>_<p>.csv, r03_pmc_sq_<w>_<p>.csv   L2 hit / miss, SQ wave-cycle split
usage: python scratch/r03_make_profiles.py <workload> <precision>"""
import csv, os, shutil, sys
w, p = sys.argv[1], sys.argv[2]
src = f"gpurun_out/r03prof_{w}_{p}"
tag = f"{w}_{p}"
cmd = f"python3 bench.py --workload {w} --precision {p} --no-cpu-baseline"
if os.path.exists(f"{src}/kernel_stats.csv"):
    shutil.copy(f"{src}/kernel_stats.csv", f"profiles/r03_bench_{tag}_kernel_stats.csv")
    shutil.copy(f"{src}/bench_under_rocprof.json", f"profiles/r03_bench_{tag}_under_rocprof.json")


def table(name):
    path = f"{src}/{name}_summary.csv"
    if not os.path.exists(path):
        return None
    lines = [ln.rstrip("\n") for ln in open(path) if ln.strip()]
    cols = lines[0].split(",")
    out = {}
    for ln in lines[1:]:                       # kernel names contain commas: split from the right
        parts = ln.rsplit(",", len(cols) - 1)
        out[parts[0].strip('"')] = dict(zip(cols, [parts[0].strip('"')] + parts[1:]))
    return out


f, wr = table("fetch"), table("write")
if f and wr:
    with open(f"profiles/r03_pmc_hbm_traffic_{tag}.csv", "w") as out:
        out.write(f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- {cmd} --steps 3 --warmup 2 --no-graph\n"
                  "# raw counter values in KB per launch; on gfx950 FETCH_SIZE reads 1/2 of wide coalesced streams "
                  "(MI355X_MICROARCH.md HBM): hbm_read ~= 2*FETCH_SIZE\n")
        out.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
        for k, r in f.items():
            out.write(f"\"void ali::{k}\",{r['launches']},{float(r['FETCH_SIZE']):.1f},{float(wr.get(k, {'WRITE_SIZE': 0})['WRITE_SIZE']):.1f}\n")
for name, hdr in (("tcc", "TCC_HIT_sum TCC_MISS_sum"), ("sq", "SQ_* (quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES in cycles)")):
    path = f"{src}/{name}_summary.csv"
    if os.path.exists(path):
        with open(f"profiles/r03_pmc_{name}_{tag}.csv", "w") as out:
            out.write(f"# rocprofv3 --kernel-trace --pmc {hdr} -- {cmd} --steps 3 --warmup 2 --no-graph; mean per launch\n")
            lines = [ln.rstrip("\n") for ln in open(path) if ln.strip()]
            n = len(lines[0].split(","))
            out.write(lines[0] + "\n")
            for ln in lines[1:]:
                parts = ln.rsplit(",", n - 1)
                out.write("\"" + parts[0].strip('"') + "\"," + ",".join(parts[1:]) + "\n")
print(sorted(x for x in os.listdir("profiles") if x.startswith("r03") and tag in x))
