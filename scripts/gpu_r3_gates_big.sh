#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for r in 100000 0 100000 0; do echo "== headline GATES_ROWS=$r"; LRAM_SLSTM_GATES_ROWS=$r run --steps 40 --warmup 8; done
for r in 100000 0; do echo "== 16M B=1024 GATES_ROWS=$r"; LRAM_SLSTM_GATES_ROWS=$r run --batch 1024 --steps 48 --warmup 8; done
for r in 100000 0; do echo "== 16M B=256 GATES_ROWS=$r"; LRAM_SLSTM_GATES_ROWS=$r run --batch 256 --steps 100 --warmup 10; done
for r in 100000 0; do echo "== 206M B=512 GATES_ROWS=$r"; LRAM_SLSTM_GATES_ROWS=$r run --config xlstm_206m --batch 512 --steps 16 --warmup 3; done
