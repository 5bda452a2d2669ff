#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm" 2>&1 | tail -3
for BMV in 0 128; do echo "LRAM_GEMM_BM=$BMV"; LRAM_GEMM_BM=$BMV bash scripts/gpu_gemm.sh bf16x3 2>/dev/null | grep -E "down|out|head|ffn|16m_up " ; done
run() { python bench.py --no-cpu-baseline --no-stream-ceilings --host-io-steps 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL $*', '->', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3),'ms/step')"; }
for BMV in 0 128 0 128; do export LRAM_GEMM_BM=$BMV; LABEL="bm=$BMV"
run --steps 40 --warmup 8
run --config mamba_48m --batch 2048 --steps 32 --warmup 4
run --config xlstm_206m --batch 512 --steps 16 --warmup 2
done
