#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --no-cpu-baseline --no-stream-ceilings --host-io-steps 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL $*', '->', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3),'ms/step | pass', r.get('state_pass_avg_ms') or r.get('avg_launch_ms'))"; }
for S in 1 0; do export LRAM_SPLIT_UP=$S; LABEL="split=$S"
run --config xlstm_206m --batch 512 --steps 16 --warmup 2
run --config xlstm_206m --batch 512 --steps 16 --warmup 2 --state eager
run --config xlstm_16m --batch 32 --steps 100 --warmup 10
run --config xlstm_16m --batch 1 --steps 100 --warmup 10
run --config xlstm_c1 --batch 32 --steps 200 --warmup 20
run --config xlstm_16m --batch 1024 --steps 32 --warmup 4
done
