#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_realbatch.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -m gpu -x > $OUT/pytest_fold.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_fold.log
bash scripts/gpu_ab.sh "LRAM_FOLD_FUSED=0" "LRAM_FOLD_FUSED=1"; cp $OUT/ab.txt $OUT/ab_fold.txt
BENCH_ARGS="--config xlstm_206m --batch 512" bash scripts/gpu_ab.sh "LRAM_FOLD_FUSED=0" "LRAM_FOLD_FUSED=1"; cp $OUT/ab.txt $OUT/ab_fold_206m.txt
BENCH_ARGS="--batch 1024" bash scripts/gpu_ab.sh "LRAM_FOLD_FUSED=0" "LRAM_FOLD_FUSED=1"; cp $OUT/ab.txt $OUT/ab_fold_1024.txt
