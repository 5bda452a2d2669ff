#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_compat.py tests/test_gpu_parity.py -q -m gpu -x -k "compat or mamba" > $OUT/pytest_compat.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_compat.log
BENCH_ARGS="--config mamba_48m --batch 2048 --mamba-compat --env-act-dim 4" bash scripts/gpu_ab.sh "LRAM_COMPAT_SHARE=0 LRAM_GEMM=bf16x3" "LRAM_COMPAT_SHARE=0" "LRAM_COMPAT_SHARE=1"
