#!/bin/bash
# A/B of dt_proj inside the selective-state-update kernel (LRAM_MAMBA_DT_FUSE), same box
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compat.py tests/test_gpu_edge.py -q -m gpu -k "mamba or compat or prefill" 2>&1 | tail -4
for rep in 1 2; do
for f in 1 0; do
  echo "== LRAM_MAMBA_DT_FUSE=$f rep $rep"
  LRAM_MAMBA_DT_FUSE=$f python bench.py --config mamba_48m --batch 2048 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
done
for f in 1 0; do
  echo "== compat LRAM_MAMBA_DT_FUSE=$f"
  LRAM_MAMBA_DT_FUSE=$f python bench.py --config mamba_48m --batch 2048 --steps 12 --warmup 4 --no-cpu-baseline --mamba-compat --env-act-dim 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
