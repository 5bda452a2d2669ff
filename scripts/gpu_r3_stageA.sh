#!/bin/bash
# round-3 stage A: new GPU tests, GN_FUSE A/B on the headline, MFMA-busy counters + durations per projection shape
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_wrapper_trace.py tests/test_backbone_golden.py tests/test_gpu_persistent.py tests/test_gpu_lazy.py -q -m gpu -x > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_new.log
LRAM_LAZY_KPREFETCH=0 timeout 600 python -m pytest tests/test_gpu_lazy.py -q -m gpu -x > $OUT/pytest_kpre0.log 2>&1; echo "pytest KPREFETCH=0 rc=$?"; tail -3 $OUT/pytest_kpre0.log
bash scripts/gpu_ab.sh "LRAM_GN_FUSE=0" "LRAM_GN_FUSE=1"
bash scripts/pmc_gemm.sh bf16x3 f32 > $OUT/pmc_gemm.txt 2>&1; cat $OUT/pmc_gemm.txt
bash scripts/gpu_gemm.sh bf16x3 f32 > $OUT/gemm_us.txt 2>&1; cat $OUT/gemm_us.txt
