#!/bin/bash
# kernel stats of the chunkwise prefill (206M, 64 envs x 512 timesteps; and 16M, 1024 x 63)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for cfg in "xlstm_206m 64 512" "xlstm_16m 1024 63"; do
  tag=$(echo $cfg | tr ' ' '_'); rm -rf $OUT/prof_pre_$tag
  PREFILL_MODES=chunkwise rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_pre_$tag -- python3 $R/scripts/bench_prefill.py $cfg > $OUT/prefill_$tag.log 2>&1
  cat $OUT/prefill_$tag.log | tail -1
  f=$(find $OUT/prof_pre_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[:14]:
    print(r[0][:90].ljust(90), *r[1:5])
PY
  cp $f $OUT/prefill_kernel_stats_$tag.csv; rm -rf $OUT/prof_pre_$tag
done
