#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 32 64; do for wk in 0 1000000 3000000 5000000 8000000 13000000; do echo "== mamba48m B=$b WORK=$wk"; LRAM_GEMM_SKINNY_WORK=$wk run --config mamba_48m --batch $b --steps 150 --warmup 10; done; done
for b in 64 128; do for wk in 0 4000000 7000000 13000000 26000000; do echo "== 16M B=$b WORK=$wk"; LRAM_GEMM_SKINNY_WORK=$wk run --batch $b --steps 150 --warmup 10; done; done
for b in 64; do for wk in 0 2000000 13000000 26000000 60000000; do echo "== 206M B=$b WORK=$wk"; LRAM_GEMM_SKINNY_WORK=$wk run --config xlstm_206m --batch $b --steps 40 --warmup 5; done; done
