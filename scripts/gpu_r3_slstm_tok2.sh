#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 1 2 8; do for m in 1 9; do echo "== 16M B=$b LRAM_SLSTM_FUSED_MIN=$m"; LRAM_SLSTM_FUSED_MIN=$m run --batch $b --steps 400 --warmup 40; done; done
for m in 1 9; do echo "== 206M B=1 LRAM_SLSTM_FUSED_MIN=$m"; LRAM_SLSTM_FUSED_MIN=$m run --config xlstm_206m --batch 1 --steps 100 --warmup 10; done
for r in 4096 512; do echo "== 16M B=2048 (1024-env slices) LRAM_SLSTM_FUSED_ROWS=$r"; LRAM_SLSTM_FUSED_ROWS=$r run --batch 2048 --steps 40 --warmup 5; done
for r in 4096 512; do echo "== 16M B=4096 LRAM_SLSTM_FUSED_ROWS=$r"; LRAM_SLSTM_FUSED_ROWS=$r run --batch 4096 --steps 40 --warmup 5; done
for r in 512 0; do echo "== 206M B=512 LRAM_SLSTM_FUSED_ROWS=$r"; LRAM_SLSTM_FUSED_ROWS=$r run --config xlstm_206m --batch 512 --steps 16 --warmup 3; done
for r in 512 0; do echo "== 16M B=1024 LRAM_SLSTM_FUSED_ROWS=$r"; LRAM_SLSTM_FUSED_ROWS=$r run --batch 1024 --steps 48 --warmup 5; done
