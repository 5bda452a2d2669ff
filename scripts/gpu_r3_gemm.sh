#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for pc in 0 1; do
export LRAM_F16_PC=$pc
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm_f16x2" > $OUT/pytest_gemm_pc$pc.log 2>&1; echo "pytest pc=$pc rc=$?"; tail -3 $OUT/pytest_gemm_pc$pc.log
bash scripts/gpu_gemm.sh f16x2 > $OUT/gemm_us_pc$pc.txt 2>&1; echo "PC=$pc"; cat $OUT/gemm_us_pc$pc.txt
done
