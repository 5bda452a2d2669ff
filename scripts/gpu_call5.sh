#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
bash scripts/gpu_prof.sh r02a --steps 16 --warmup 4 --host-io-steps 0 --no-stream-ceilings | head -20
f=$(find $OUT/prof_r02a -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $f -3 30 -v > $OUT/timeline_r02a.txt; head -32 $OUT/timeline_r02a.txt
