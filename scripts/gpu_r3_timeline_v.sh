#!/bin/bash
# verbose per-kernel timeline of one steady-state env-step -> gpurun_out/timeline_v_<tag>.txt ; usage: <tag> <bench args>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd $R; TAG=$1; shift
bash scripts/gpu_prof.sh tlv "$@" --no-kernel-timing > /dev/null 2>&1
t=$(find $OUT/prof_tlv -name "*kernel_trace.csv" | head -1); python scripts/timeline.py $t -3 30 -v > $OUT/timeline_v_$TAG.txt
rm -rf $OUT/prof_tlv
head -5 $OUT/timeline_v_$TAG.txt
