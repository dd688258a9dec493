#!/bin/bash
# the three profiled workloads of round 3, all passes; then the seven bench lines
cd "$GRAFT_REPO_ROOT"
bash scratch/r03_collect.sh mnist f32 40 > gpurun_out/r03_collect_mnist.log 2>&1 && echo "mnist done" &&
bash scratch/r03_collect.sh audio f32 10 > gpurun_out/r03_collect_audio.log 2>&1 && echo "audio done" &&
bash scratch/r03_collect.sh esrf f16 8 > gpurun_out/r03_collect_esrf_f16.log 2>&1 && echo "esrf done" &&
bash scratch/r03_bench_all.sh
