#!/bin/bash
# usage: gpu_gemm.sh [kernels...]  -> per-shape GEMM kernel durations (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_gemm; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/bench_gemm.py "$@" > $OUT/order.json 2> $OUT/err.log
python3 $R/scripts/parse_gemm_trace.py $OUT
