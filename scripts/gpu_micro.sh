#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-300
for m in 0 1; do
python bench.py --steps 32 --warmup 4 --no-cpu-baseline --micro $m 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('micro $m', round(d['value']), 'env-steps/s', round(d['ms_per_step'],2),'ms  cell', round(r['avg_launch_ms'],3),'ms', round(r['achieved']),'GB/s share', round(r['kernel_share_of_step'],3))"
done
python bench.py --config mamba_48m --batch 2048 --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-140
