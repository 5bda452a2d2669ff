#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3),'ms/step')"; }
run --config xlstm_16m --batch 1 --steps 200 --warmup 20
run --config xlstm_16m --batch 2 --steps 200 --warmup 20
run --config xlstm_16m --batch 32 --steps 200 --warmup 20
run --config mamba_48m --batch 1 --steps 200 --warmup 20
run --config xlstm_c1 --batch 32 --steps 300 --warmup 30
run --config xlstm_206m --batch 1 --steps 100 --warmup 10
