#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_persistent.py -q -m gpu -x 2>&1 | tail -25
run() { timeout 120 python bench.py --no-cpu-baseline --no-stream-ceilings --no-kernel-timing "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$LABEL $*', '->', round(d['value']), 'env-steps/s', round(d['ms_per_step'],4),'ms/step host-inclusive', round(d.get('host_io',{}).get('ms_per_step',0),4))"; }
for W in 32 64 128 256; do export LRAM_PERSIST_WGS=$W; LABEL="wgs=$W"; run --config xlstm_16m --batch 1 --steps 200 --warmup 20; done
export LRAM_PERSIST_WGS=128; LABEL="wgs=128"
run --config xlstm_16m --batch 8 --steps 200 --warmup 20
run --config xlstm_c1 --batch 8 --steps 200 --warmup 20
run --config xlstm_206m --batch 1 --steps 100 --warmup 10
export LRAM_PERSISTENT=0; LABEL="launch path"
run --config xlstm_16m --batch 1 --steps 200 --warmup 20
run --config xlstm_16m --batch 8 --steps 200 --warmup 20
run --config xlstm_206m --batch 1 --steps 100 --warmup 10
