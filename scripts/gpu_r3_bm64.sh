#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 128 192 256 341 512; do for r in 1024 0; do echo "== 16M B=$b BM64_ROWS=$r"; LRAM_BF16_BM64_ROWS=$r run --batch $b --steps 100 --warmup 10; done; done
for b in 64 128 512; do for r in 1024 0; do echo "== 206M B=$b BM64_ROWS=$r"; LRAM_BF16_BM64_ROWS=$r run --config xlstm_206m --batch $b --steps 30 --warmup 5; done; done
for b in 64 256; do for r in 1024 0; do echo "== mamba B=$b BM64_ROWS=$r"; LRAM_BF16_BM64_ROWS=$r run --config mamba_48m --batch $b --steps 100 --warmup 10; done; done
