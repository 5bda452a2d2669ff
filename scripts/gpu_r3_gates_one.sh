#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 8 32 128; do for f in 1 0 1 0; do echo "== 16M B=$b GATES_ONE=$f"; LRAM_SLSTM_GATES_ONE=$f run --batch $b --steps 300 --warmup 30; done; done
for b in 16 64; do for f in 1 0; do echo "== 206M B=$b GATES_ONE=$f"; LRAM_SLSTM_GATES_ONE=$f run --config xlstm_206m --batch $b --steps 60 --warmup 10; done; done
