#!/bin/bash
# f16x2 in the engine: parity suite, then A/B of the projection kernel on the headline and on Mamba-48M
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
bash scripts/gpu_ab.sh "LRAM_GEMM=bf16x3" "LRAM_GEMM=f16x2"; cp $OUT/ab.txt $OUT/ab_headline.txt
BENCH_ARGS="--config mamba_48m --batch 2048" bash scripts/gpu_ab.sh "LRAM_GEMM=bf16x3" "LRAM_GEMM=f16x2"; cp $OUT/ab.txt $OUT/ab_mamba.txt
BENCH_ARGS="--config xlstm_206m --batch 512" bash scripts/gpu_ab.sh "LRAM_GEMM=bf16x3" "LRAM_GEMM=f16x2"; cp $OUT/ab.txt $OUT/ab_206m.txt
