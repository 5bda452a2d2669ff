#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for db in 0 1; do
export LRAM_F16_DB=$db
bash scripts/gpu_gemm.sh f16x2 > $OUT/gemm_us_db$db.txt 2>&1; echo "DB=$db"; grep -E "_s |16m_head|206m_down|mamba_out " $OUT/gemm_us_db$db.txt
done
export LRAM_F16_DB=0
for bm in 64 128; do LRAM_GEMM_BM=$bm bash scripts/gpu_gemm.sh f16x2 > $OUT/gemm_us_bm$bm.txt 2>&1; echo "DB=0 BM=$bm"; grep -E "_s " $OUT/gemm_us_bm$bm.txt; done
export LRAM_F16_DB=1
for bm in 64 128; do LRAM_GEMM_BM=$bm bash scripts/gpu_gemm.sh f16x2 > $OUT/gemm_us_bm$bm.txt 2>&1; echo "DB=1 BM=$bm"; grep -E "_s " $OUT/gemm_us_bm$bm.txt; done
