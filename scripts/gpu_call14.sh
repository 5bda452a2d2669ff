#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --no-cpu-baseline --no-stream-ceilings --host-io-steps 0 --steps 40 --warmup 8 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL $*', '->', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3),'ms/step | pass', round(r.get('state_pass_avg_ms') or 0,3), 'fold', round(r.get('fold_avg_ms') or 0,3))"; }
run --micro 2
run --micro 3
run --micro 4
LRAM_LAZY_UNROLL=8 run --micro 3
GPU_MAX_HW_QUEUES=8 run --micro 3
GPU_MAX_HW_QUEUES=8 run --micro 4
run --micro 2
