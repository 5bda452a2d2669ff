#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "few_row or xlstm_16m_shapes or c1_b32 or mamba_48m_shapes or slstm_token" 2>&1 | tail -4
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 4 12 32 64 128 256; do for r in 100000 0; do echo "== 16M B=$b LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --batch $b --steps 150 --warmup 20; done; done
for r in 100000 0; do echo "== C1 B=32 LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --config xlstm_c1 --batch 32 --steps 300 --warmup 30; done
for b in 16 64; do for r in 100000 0; do echo "== 206M B=$b LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --config xlstm_206m --batch $b --steps 40 --warmup 5; done; done
for b in 16 64; do for r in 100000 0; do echo "== mamba48m B=$b LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --config mamba_48m --batch $b --steps 100 --warmup 10; done; done
