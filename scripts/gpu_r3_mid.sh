#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 192 256 512; do for r in 384 768 1536; do echo "== 16M B=$b SKINNY_ROWS=$r"; LRAM_GEMM_SKINNY_ROWS=$r run --batch $b --steps 100 --warmup 10; done; done
