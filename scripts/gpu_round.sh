#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel-trace summary of the same bench command, PMC pass.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log | cut -c1-200
python bench.py --steps ${STEPS:-64} --warmup 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cut -c1-400 $OUT/bench.json; tail -2 $OUT/bench.err | cut -c1-200
bash scripts/gpu_prof.sh headline --steps 16 --warmup 4
bash scripts/pmc_pass.sh 2>&1 | grep -E "rc=|hbm_bytes_per_env|fetch_reported|write_reported"
