#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for cfg in "84 16" "56 16" "56 8" "40 8" "84 8"; do
  set -- $cfg
  LRAM_CELL_LDS_PAD_KB=$1 LRAM_CELL_UNROLL=$2 python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pad $1 unroll $2', round(d['value']), 'env-steps/s', round(d['ms_per_step'],2),'ms  cell', round(r['avg_launch_ms'],3),'ms', round(r['achieved']),'GB/s share', round(r['kernel_share_of_step'],3), 'standalone', round(r['standalone']['achieved']))"
done
