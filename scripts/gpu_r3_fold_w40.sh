#!/bin/bash
# fold kernel with 40 window rows in LDS (four workgroups per CU) vs 48 (three); variants with 128 / 256 rows of C per workgroup
run() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],3), 'read', r.get('avg_launch_ms'), 'fold', r.get('fold_avg_ms'))"; }
for rep in 1 2; do
echo "== W40=1"; LRAM_FOLD_W40=1 run
echo "== W40=0"; LRAM_FOLD_W40=0 run
done
echo "== kfr128 (W40=0)"; LRAM_FOLD_W40=0 LRAM_LIB_VARIANT=kfr128 run
echo "== kfr256 (W40=0)"; LRAM_FOLD_W40=0 LRAM_LIB_VARIANT=kfr256 run
echo "== 206M W40=1"; LRAM_FOLD_W40=1 run --config xlstm_206m --batch 512 --steps 16 --warmup 3
echo "== 206M W40=0"; LRAM_FOLD_W40=0 run --config xlstm_206m --batch 512 --steps 16 --warmup 3
