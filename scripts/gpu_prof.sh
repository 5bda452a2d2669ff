#!/bin/bash
# usage: gpu_prof.sh <tag> <bench args...>   -> compact kernel stats (top 14) + copies under gpurun_out/prof_<tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[:15]:
    print(r[0][:70].ljust(70), *r[1:5])
PY
cut -c1-160 $OUT/bench.json
