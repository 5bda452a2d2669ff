#!/bin/bash
# Mamba-48M at 2048 slots: number of free-running env slices
for m in 2 3 4 2 3 4; do
  echo "== micro $m"
  python bench.py --config mamba_48m --batch 2048 --steps 30 --warmup 8 --no-cpu-baseline --host-io-steps 0 --micro $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
