#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 1 8 32 128 256 512; do for r in 384 0; do echo "== 16M B=$b LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --batch $b --steps 150 --warmup 20; done; done
for b in 16 32 64; do for r in 384 0; do echo "== 206M B=$b LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --config xlstm_206m --batch $b --steps 40 --warmup 5; done; done
for b in 16 32 64; do for r in 384 0; do echo "== mamba48m B=$b LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --config mamba_48m --batch $b --steps 100 --warmup 10; done; done
for r in 384 0; do echo "== prefill 206M 64x512 LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r PREFILL_MODES=chunkwise python scripts/bench_prefill.py xlstm_206m 64 512 | tail -1; done
for r in 384 0; do echo "== headline LRAM_GEMM_SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --steps 40 --warmup 8; done
