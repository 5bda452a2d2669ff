#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python scripts/debug_depth.py 2>&1 | grep -v amdgpu.ids | tee $OUT/debug_depth.txt
echo "--- f32 gemm"; LRAM_GEMM=f32 python scripts/debug_depth.py 8 20 2>&1 | grep -v amdgpu.ids | tee -a $OUT/debug_depth.txt
for v in 4 5 6; do LRAM_COPY_VARIANT=$v python scripts/bench_streams.py 2>/dev/null; done | tee $OUT/streams2.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "c5 or long_runs" 2>&1 | tail -15
