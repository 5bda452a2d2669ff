#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm" > $OUT/pytest_gemm.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gemm.log
bash scripts/gpu_gemm.sh f16x2 bf16x3 > $OUT/gemm_us.txt 2>&1; cat $OUT/gemm_us.txt
bash scripts/pmc_gemm.sh f16x2 > $OUT/pmc_gemm.txt 2>&1; cat $OUT/pmc_gemm.txt
