#!/bin/bash
export LRAM_STATE=lazy LRAM_CELL_LDS_PAD_KB=0
for m in 2 3 4; do
  python bench.py --steps 64 --warmup 16 --no-cpu-baseline --micro $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('lazy micro $m:', round(d['value']), round(d['ms_per_step'],3), 'cell avg ms', r.get('avg_launch_ms'))"
done
