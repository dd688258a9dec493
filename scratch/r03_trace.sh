#!/bin/bash
# kernel trace of the default bench (graph replay) -> per-iteration timeline
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03trace
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1
grep '"metric"' $OUT/bench.log | tail -1 > $OUT/bench.json
python scratch/iter_timeline.py $OUT/ks/ks_kernel_trace.csv 20 > $OUT/timeline.txt
find $OUT/ks -name "*kernel_trace.csv" -delete || true
tail -45 $OUT/timeline.txt
