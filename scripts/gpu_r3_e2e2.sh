#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -x -k "mamba or gemm or xlstm_tiny or xlstm_16m or c1 or compat or checkpoint" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
bash scripts/gpu_ab.sh "LRAM_GEMM=bf16x3" "LRAM_GEMM=f16x2"; cp $OUT/ab.txt $OUT/ab_headline.txt
BENCH_ARGS="--config mamba_48m --batch 2048" bash scripts/gpu_ab.sh "LRAM_GEMM=bf16x3" "LRAM_GEMM=f16x2"; cp $OUT/ab.txt $OUT/ab_mamba.txt
