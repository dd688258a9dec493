#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats of the default bench command + separate PMC passes (HBM traffic).
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r01prof
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python bench.py > $OUT/bench_under_rocprof.log 2>&1
grep '"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
# keep the merged-back payload small: drop the per-dispatch traces of the stats run
find $OUT/ks -name "*kernel_trace.csv" -size +20M -delete || true
