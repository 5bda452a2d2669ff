#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 4 12 32 64 128 256; do for r in 384 0; do echo "== 16M B=$b ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --batch $b --steps 150 --warmup 20; done; done
for r in 384 0; do echo "== C1 B=32 ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --config xlstm_c1 --batch 32 --steps 300 --warmup 30; done
for b in 16 64; do for r in 384 0; do echo "== 206M B=$b ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --config xlstm_206m --batch $b --steps 40 --warmup 5; done; done
for b in 16 32 64; do for r in 384 0; do echo "== mamba48m B=$b ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --config mamba_48m --batch $b --steps 150 --warmup 10; done; done
