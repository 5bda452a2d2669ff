#!/bin/bash
# usage: gpu_gemm_panel.sh [kernels...] -> GEMM durations per tile-order panel width (LRAM_GEMM_PANEL; "" = the 2-D XCD split)
R=${GRAFT_REPO_ROOT:-/root/repo}
for pw in "" 2 4 6 8 10 13 16 20 99; do
  echo "=== LRAM_GEMM_PANEL=${pw:-unset}"
  if [ -z "$pw" ]; then unset LRAM_GEMM_PANEL; else export LRAM_GEMM_PANEL=$pw; fi
  bash $R/scripts/gpu_gemm.sh "$@" 2>&1 | grep -E "16m_up |16m_down |mamba_in |mamba_out |206m_up |206m_down |c5_up|c5_down|prefill_up|206m_up_s|16m_head"
done
