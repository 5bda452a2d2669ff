#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "few_row or xlstm_16m_shapes or c1_b32" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 12 32 64 128; do for f in 1 0; do echo "== 16M B=$b FORM=$f"; LRAM_GEMM_SKINNY_FORM=$f run --batch $b --steps 150 --warmup 20; done; done
for f in 1 0; do echo "== C1 B=32 FORM=$f"; LRAM_GEMM_SKINNY_FORM=$f run --config xlstm_c1 --batch 32 --steps 300 --warmup 30; done
for f in 1 0; do echo "== 206M B=16 FORM=$f"; LRAM_GEMM_SKINNY_FORM=$f run --config xlstm_206m --batch 16 --steps 40 --warmup 5; done
for f in 1 0; do echo "== mamba B=16 FORM=$f"; LRAM_GEMM_SKINNY_FORM=$f run --config mamba_48m --batch 16 --steps 100 --warmup 10; done
