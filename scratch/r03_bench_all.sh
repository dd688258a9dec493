#!/bin/bash
# all seven bench lines (one per workload x precision), printed as "workload precision ms img/s"
cd "$GRAFT_REPO_ROOT"
for wp in "esrf f32" "esrf f16" "whale f32" "whale f16" "audio f32" "audio f16" "mnist f32"; do
  set -- $wp
  [[ -n "$ONLY" && "$ONLY" != *"$2"* ]] && continue
  python bench.py --workload $1 --precision $2 --no-cpu-baseline ${STEPS:+--steps $STEPS} > gpurun_out/r03_bench_$1_$2.json 2> gpurun_out/r03_bench_$1_$2.err || { echo "$1 $2 FAILED"; tail -5 gpurun_out/r03_bench_$1_$2.err; continue; }
  python - "$1" "$2" <<PY
import json,sys
d=json.loads([l for l in open("gpurun_out/r03_bench_%s_%s.json"%(sys.argv[1],sys.argv[2])) if l.startswith("{")][-1])
print(sys.argv[1],sys.argv[2],d["ms_per_step"],round(d["value"],1),"frac",d["roofline"]["frac"])
PY
done
