#!/bin/bash
# tests + headline bench + rocprof + PMC in both state modes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log | cut -c1-200
python bench.py --steps 64 --warmup 16 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
python bench.py --steps 64 --warmup 16 --state eager --no-cpu-baseline > $OUT/bench_eager.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_eager.json
bash scripts/gpu_prof.sh headline --steps 16 --warmup 4 | head -12
bash scripts/pmc_pass.sh 2>&1 | grep -E "rc=|hbm_bytes_per_env|state_mode"
BENCH_ARGS="--state eager" bash scripts/pmc_pass.sh 2>&1 | grep -E "rc=|hbm_bytes_per_env|state_mode"
