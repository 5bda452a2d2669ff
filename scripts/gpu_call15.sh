#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_chunk.py -q -m gpu -x 2>&1 | tail -5
run() { python bench.py --no-cpu-baseline --no-stream-ceilings --host-io-steps 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL $*', '->', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3),'ms/step')"; }
for A in 1 0 1 0; do export LRAM_GEMM_A3=$A; LABEL="a3=$A"
run --steps 40 --warmup 8
run --config mamba_48m --batch 2048 --steps 32 --warmup 4
run --config xlstm_206m --batch 512 --steps 16 --warmup 2
done
