#!/bin/bash
# kernel stats + per-kernel timeline of one steady-state step of the headline (or BENCH_ARGS) under rocprofv3
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
TAG=${1:-headline}; shift
bash scripts/gpu_prof.sh $TAG --steps 16 --warmup 4 "$@" | head -24
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $t -3 30 -v > gpurun_out/timeline_$TAG.txt; head -30 gpurun_out/timeline_$TAG.txt
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/kernel_stats_$TAG.csv
