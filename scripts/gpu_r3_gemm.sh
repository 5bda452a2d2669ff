#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for db in 0 1; do
export LRAM_F16_DB=$db
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm_f16x2" > $OUT/pytest_gemm_db$db.log 2>&1; echo "pytest db=$db rc=$?"; tail -3 $OUT/pytest_gemm_db$db.log
bash scripts/gpu_gemm.sh f16x2 > $OUT/gemm_us_db$db.txt 2>&1; echo "DB=$db"; cat $OUT/gemm_us_db$db.txt
done
