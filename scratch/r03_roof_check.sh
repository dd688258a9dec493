#!/bin/bash
# r03: bench.py's live roofline (HIP events, in sequence) against rocprofv3's kernel trace of the SAME process.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03roof
rm -rf $OUT && mkdir -p $OUT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_plain.log 2>&1
grep '"metric"' $OUT/bench_plain.log | tail -1 > $OUT/bench_plain.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
grep '"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
find $OUT/ks -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python scratch/kstats.py $OUT/kernel_stats.csv 48 40 > $OUT/kstats.txt
find $OUT/ks -name "*kernel_trace.csv" -size +40M -delete || true
cat $OUT/bench_plain.json; echo; cat $OUT/bench_under_rocprof.json; echo; head -30 $OUT/kstats.txt
