"""Turn the rocprofv3 outputs of a round (gpurun_out/...) into the small tracked summaries under profiles/.

  python scratch/make_profiles.py r02 <kernel_stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> \
         <sq_counter_collection.csv> <tag>
"""
import csv, re, sys
from collections import defaultdict
rnd, stats, fetch, write, sq, tag = sys.argv[1:7]


def per_kernel(path, counters=None):
    acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"])
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[n][r["Counter_Name"]] += 1
    return acc, cnt


# 1. kernel stats: copy as is (already a summary)
open(f"profiles/{rnd}_bench_{tag}_kernel_stats.csv", "w").write(open(stats).read())
# 2. HBM traffic: FETCH_SIZE / WRITE_SIZE, KB per launch
fa, fc = per_kernel(fetch)
wa, wc = per_kernel(write)
with open(f"profiles/{rnd}_pmc_hbm_traffic_{tag}.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 "
            "--warmup 2 --no-cpu-baseline\n# raw counter values in KB per launch; on gfx950 FETCH_SIZE reads 1/2 of wide "
            "coalesced streams (MI355X_MICROARCH.md HBM): hbm_read ~= 2*FETCH_SIZE\n")
    f.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
    for k in sorted(fa, key=lambda k: -fa[k]["FETCH_SIZE"]):
        n = fc[k]["FETCH_SIZE"]
        if n < 4:
            continue
        w = wa[k]["WRITE_SIZE"] / max(wc[k]["WRITE_SIZE"], 1)
        f.write(f"\"{k}\",{n},{fa[k]['FETCH_SIZE'] / n:.1f},{w:.1f}\n")
# 3. SQ pass: per launch means + derived shares
sa, sc = per_kernel(sq)
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
         "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32"]
with open(f"profiles/{rnd}_pmc_sq_{tag}.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc " + " ".join(names) + " -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline\n"
            "# per-launch means.  SQ_WAVE_CYCLES = WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA "
            "pipe busy, dependencies) + ACTIVE_INST_ANY (issuing); these count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts "
            "cycles summed over SIMDs (MI355X_MICROARCH.md).  mfma_busy = MFMA_BUSY_CYCLES / (duration * clock * 1024 SIMDs) "
            "is computed in DESIGN.md from the kernel-trace durations.\n")
    f.write("kernel,launches," + ",".join(names) + ",wait_share,issue_stall_share,active_share\n")
    for k in sorted(sa, key=lambda k: -sa[k]["SQ_WAVE_CYCLES"]):
        n = max(sc[k].values())
        if n < 4:
            continue
        v = {c: sa[k][c] / max(sc[k][c], 1) for c in names}
        wc_ = max(v["SQ_WAVE_CYCLES"], 1.0)
        f.write(f"\"{k}\",{n}," + ",".join(f"{v[c]:.4g}" for c in names)
                + f",{v['SQ_WAIT_ANY'] / wc_:.3f},{v['SQ_WAIT_INST_ANY'] / wc_:.3f},{v['SQ_ACTIVE_INST_ANY'] / wc_:.3f}\n")
print("wrote profiles/", rnd, tag)
