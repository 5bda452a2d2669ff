#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for abl in 0 1 2 3 4 5 6; do
export LRAM_F16_ABL=$abl LRAM_GEMM_BM=128
bash scripts/gpu_gemm.sh f16x2 > $OUT/gemm_abl$abl.txt 2>&1; echo "ABL=$abl"; grep -E "16m_up |mamba_in|mamba_out|16m_down" $OUT/gemm_abl$abl.txt
done
