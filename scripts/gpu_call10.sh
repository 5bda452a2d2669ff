#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --no-cpu-baseline --no-stream-ceilings --host-io-steps 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$*', '->', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3),'ms/step | pass', r.get('state_pass_avg_ms') or r.get('avg_launch_ms'))"; }
run --config xlstm_206m --batch 512 --steps 16 --warmup 2
run --config xlstm_16m --batch 32 --steps 100 --warmup 10
run --config xlstm_16m --batch 1024 --steps 32 --warmup 4
run --config xlstm_16m --steps 40 --warmup 8
timeout 900 python -m pytest tests/test_gpu_lazy.py -q -m gpu -x 2>&1 | tail -3
