#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 256 512; do for t in 128 64 24 0; do echo "== 16M B=$b SPLITK_TILES=$t"; LRAM_SPLITK_TILES=$t run --batch $b --steps 100 --warmup 10; done; done
for b in 256 512; do for t in 64; do echo "== 16M B=$b GEMM_BM=$t"; LRAM_GEMM_BM=$t run --batch $b --steps 100 --warmup 10; done; done
