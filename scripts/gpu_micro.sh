#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3; do
python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],2),'ms  cell', round(r['avg_launch_ms'],3),'ms', round(r['achieved']),'GB/s share', round(r['kernel_share_of_step'],3), 'standalone', round(r['standalone']['achieved']))"
done
