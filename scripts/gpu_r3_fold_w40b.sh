#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],3), 'read', r.get('avg_launch_ms'), 'fold', r.get('fold_avg_ms'))"; }
for rep in 1 2 3; do for w in 1 0; do echo "== 206M W40=$w"; LRAM_FOLD_W40=$w run --config xlstm_206m --batch 512 --steps 24 --warmup 4; done; done
for rep in 1 2 3; do for w in 1 0; do echo "== 16M W40=$w"; LRAM_FOLD_W40=$w run --steps 48 --warmup 8; done; done
for rep in 1 2; do for w in 1 0; do echo "== 16M b1024 W40=$w"; LRAM_FOLD_W40=$w run --batch 1024 --steps 48 --warmup 8; done; done
