#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -45 $OUT/pytest_gpu.log | cut -c1-220
for v in 0 1 2 3 4; do LRAM_COPY_VARIANT=$v python scripts/bench_streams.py 2>/dev/null; done | tee $OUT/streams.txt
timeout 600 python bench.py --steps 64 --warmup 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json | cut -c1-3000; tail -3 $OUT/bench.err | cut -c1-300
