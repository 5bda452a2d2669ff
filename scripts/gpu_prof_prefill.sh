#!/bin/bash
# usage: gpu_prof_prefill.sh <config> <B> <L> [micro]  -> kernel stats of scripts/bench_prefill.py under rocprofv3
# (PREFILL_MODES=chunkwise restricts the run to the chunkwise mode)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_prefill_$1; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/bench_prefill.py "$@" > $OUT/out.log 2> $OUT/err.log
cat $OUT/out.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" ${TOP:-22} <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print(r[0][:86].ljust(86), *r[1:5])
PY
