#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 1 2; do for m in 1 9 1 9; do echo "== 16M B=$b MIN=$m"; LRAM_GEMM_SKINNY_MIN=$m run --batch $b --steps 400 --warmup 40; done; done
for m in 1 9; do echo "== mamba B=1 MIN=$m"; LRAM_GEMM_SKINNY_MIN=$m run --config mamba_48m --batch 1 --steps 200 --warmup 20; done
for m in 1 9; do echo "== 206M B=1 MIN=$m"; LRAM_GEMM_SKINNY_MIN=$m run --config xlstm_206m --batch 1 --steps 100 --warmup 10; done
