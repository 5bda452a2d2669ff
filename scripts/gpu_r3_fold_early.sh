#!/bin/bash
# fold kernel of compact launches: C_base tile requested together with the env's count / restart words (new) vs the build before (old)
timeout 900 python -m pytest tests/test_gpu_lazy.py -q -m gpu -x 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],4), 'fold', round(r.get('fold_avg_ms') or 0,4), 'fold alone', round((r.get('standalone') or {}).get('fold_avg_ms') or 0,4))"; }
for v in new old new old new old; do
  if [ $v = old ]; then export LRAM_LIB_VARIANT=prev; else unset LRAM_LIB_VARIANT; fi
  echo "== headline $v"; run --steps 48 --warmup 8
done
for v in new old; do
  if [ $v = old ]; then export LRAM_LIB_VARIANT=prev; else unset LRAM_LIB_VARIANT; fi
  echo "== 206M B=512 $v"; run --config xlstm_206m --batch 512 --steps 24 --warmup 4
done
