#!/bin/bash
run() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],3), 'read', r.get('avg_launch_ms'), 'fold', r.get('fold_avg_ms'))"; }
for p in 0 1 2 3 0 1; do echo "== LRAM_STREAM_PRIO=$p"; LRAM_STREAM_PRIO=$p run; done
for p in 0 1; do echo "== W40=0 PRIO=$p"; LRAM_FOLD_W40=0 LRAM_STREAM_PRIO=$p run; done
for p in 0 1; do echo "== 206M PRIO=$p"; LRAM_STREAM_PRIO=$p run --config xlstm_206m --batch 512 --steps 16 --warmup 3; done
for p in 0 1; do echo "== mamba PRIO=$p"; LRAM_STREAM_PRIO=$p run --config mamba_48m --batch 2048 --steps 30 --warmup 5; done
