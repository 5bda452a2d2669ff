#!/bin/bash
# Compact bench sweep over BASELINE.json's other configurations (parity-test cases, not the headline line).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$*', '->', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3),'ms/step', '(host-inclusive', round(d.get('host_io',{}).get('value',0)), ') |', r['kernel'], r.get('avg_launch_ms') and round(r['avg_launch_ms'],3), 'ms', r.get('achieved') and round(r['achieved']), 'GB/s | whole-step', round(d['whole_step_8d_GBps']), 'GB/s')"; }
run --config xlstm_16m --batch 1024 --steps 32 --warmup 4
run --config xlstm_c1 --batch 32 --steps 200 --warmup 20
run --config xlstm_c1 --batch 32 --steps 200 --warmup 20 --graph
run --config xlstm_16m --batch 32 --steps 100 --warmup 10
run --config xlstm_16m --batch 32 --steps 100 --warmup 10 --graph
run --config xlstm_16m --batch 1 --steps 400 --warmup 40
run --config xlstm_16m --batch 1 --steps 400 --warmup 40 --graph
run --config mamba_48m --batch 2048 --steps 32 --warmup 4
run --config mamba_48m --batch 1 --steps 100 --warmup 10 --graph
run --config xlstm_206m --batch 512 --steps 16 --warmup 2
run --config xlstm_206m --batch 512 --steps 16 --warmup 2 --micro 1
run --config xlstm_206m --batch 512 --steps 16 --warmup 2 --obs image
run --config mamba_48m --batch 2048 --steps 16 --warmup 4 --mamba-compat --env-act-dim 4
