#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for f in 1 0 1 0; do echo "== mamba B=16 FORM=$f"; LRAM_GEMM_SKINNY_FORM=$f run --config mamba_48m --batch 16 --steps 100 --warmup 10; done
for f in 1 0; do echo "== 16M B=12 FORM=$f"; LRAM_GEMM_SKINNY_FORM=$f run --batch 12 --steps 150 --warmup 20; done
for f in 1 0; do echo "== 16M B=128 FORM=$f"; LRAM_GEMM_SKINNY_FORM=$f run --batch 128 --steps 150 --warmup 20; done
