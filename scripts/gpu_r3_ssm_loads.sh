#!/bin/bash
# Mamba selective-state-update kernel with every request issued up front (new) vs the build before (csrc/_variants/prev.so)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compat.py tests/test_gpu_edge.py -q -m gpu -x -k "mamba or compat or prefill" 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],4), 'ssm', r.get('avg_launch_ms'))"; }
for v in new old new old; do
  if [ $v = old ]; then export LRAM_LIB_VARIANT=prev; else unset LRAM_LIB_VARIANT; fi
  echo "== mamba B=2048 $v"; run --config mamba_48m --batch 2048 --steps 40 --warmup 8
done
for v in new old; do
  if [ $v = old ]; then export LRAM_LIB_VARIANT=prev; else unset LRAM_LIB_VARIANT; fi
  echo "== mamba B=2048 compat $v"; run --config mamba_48m --batch 2048 --steps 12 --warmup 4 --mamba-compat --env-act-dim 4
  echo "== mamba B=64 $v"; run --config mamba_48m --batch 64 --steps 100 --warmup 10
  echo "== mamba B=1 $v"; run --config mamba_48m --batch 1 --steps 200 --warmup 20
done
