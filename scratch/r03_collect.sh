#!/bin/bash
# r03: rocprofv3 summaries of `python3 bench.py --workload W --precision P` on the GPU box (gpurun):
#   kernel-trace stats, then SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss, SQ), each with --kernel-trace only.
# usage: bash scratch/r03_collect.sh <workload> <precision> [steps] [passes: ks,fetch,write,tcc,sq]
set -e -o pipefail
W=${1:-mnist}; P=${2:-f32}; STEPS=${3:-20}; PASSES=${4:-ks,fetch,write,tcc,sq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03prof_${W}_${P}
rm -rf $OUT && mkdir -p $OUT
ARGS="--workload $W --precision $P --no-cpu-baseline"
if [[ $PASSES == *ks* ]]; then
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 bench.py $ARGS --steps $STEPS --warmup 3 > $OUT/bench_under_rocprof.log 2>&1
  grep '"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
  find $OUT/ks -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
  find $OUT/ks -name "*kernel_trace.csv" -size +30M -delete || true
fi
pmc() {  # name, counters
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python3 bench.py $ARGS --steps 3 --warmup 2 --no-graph > $OUT/$1.log 2>&1
  find $OUT/$1 -name "*counter_collection.csv" -exec python scratch/pmc_summary.py {} 3 \; > $OUT/$1_summary.csv
  rm -rf $OUT/$1
  echo "pass $1 done"
}
[[ $PASSES == *fetch* ]] && pmc fetch FETCH_SIZE
[[ $PASSES == *write* ]] && pmc write WRITE_SIZE
[[ $PASSES == *tcc* ]] && pmc tcc "TCC_HIT_sum TCC_MISS_sum"
[[ $PASSES == *sq* ]] && pmc sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"
ls -la $OUT
