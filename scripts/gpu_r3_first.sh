#!/bin/bash
# round-3 first GPU pass: full GPU suite with the per-element report, headline bench, hazard evidence
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
LRAM_TEST_REPORT=1 timeout 1500 python -m pytest tests -q -m gpu -s --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "report|passed|failed|rc=" $OUT/pytest_gpu.log | tail -30
timeout 600 python bench.py --steps 64 --warmup 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
timeout 900 python scripts/hazard_evidence.py --run $OUT/hazard.json > $OUT/hazard.log 2>&1; echo "hazard rc=$?"; tail -30 $OUT/hazard.log
