#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel-trace summary of the same bench command.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python bench.py --steps ${STEPS:-32} --warmup 4 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} sh -c 'head -25 {}'
