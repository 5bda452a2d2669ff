#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python scripts/debug_depth.py 16 20 2>&1 | grep -v amdgpu.ids | tee $OUT/debug_depth2.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "c5 or c4" 2>&1 | tail -25
