#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['config']['state_mode'])"; }
for b in 64 128 256 384; do for st in eager lazy; do echo "== 16M B=$b state=$st"; run --batch $b --steps 150 --warmup 30 --state $st; done; done
for b in 32 64; do for st in eager lazy; do echo "== 206M B=$b state=$st"; run --config xlstm_206m --batch $b --steps 50 --warmup 30 --state $st; done; done
