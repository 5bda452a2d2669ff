#!/bin/bash
# 206M at 512 env slots: the q-free folds shared out over the three sLSTM stretches (LRAM_FOLD_SPREAD) vs all behind the first
for rep in 1 2; do for f in 1 0; do
  echo "== LRAM_FOLD_SPREAD=$f"
  LRAM_FOLD_SPREAD=$f python bench.py --config xlstm_206m --batch 512 --steps 16 --warmup 3 --no-cpu-baseline --host-io-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
for f in 1 0; do
  echo "== image obs LRAM_FOLD_SPREAD=$f"
  LRAM_FOLD_SPREAD=$f python bench.py --config xlstm_206m --batch 512 --steps 16 --warmup 3 --no-cpu-baseline --host-io-steps 0 --obs image 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
for rep in 1 2; do
  echo "== headline rep $rep"
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --host-io-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
