#!/bin/bash
# the conv / gate front-end kernel without the reset flag's own round trip at its head (new build) vs the build before (variant)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -q -m gpu -x 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for v in new old new old; do
  if [ $v = old ]; then export LRAM_LIB_VARIANT=prev; else unset LRAM_LIB_VARIANT; fi
  echo "== 16M B=1 $v"; run --batch 1 --steps 400 --warmup 40
  echo "== 16M B=32 $v"; run --batch 32 --steps 300 --warmup 30
  echo "== C1 B=32 $v"; run --config xlstm_c1 --batch 32 --steps 400 --warmup 40
  echo "== headline $v"; run --steps 40 --warmup 8
done
