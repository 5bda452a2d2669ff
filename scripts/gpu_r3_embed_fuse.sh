#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_golden.py tests/test_gpu_compat.py tests/test_gpu_wrapper_trace.py -q -m gpu -x 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 1 32; do for f in 1 0 1 0; do echo "== 16M B=$b EMBED_FUSE=$f"; LRAM_EMBED_FUSE=$f run --batch $b --steps 300 --warmup 30; done; done
for f in 1 0 1 0; do echo "== C1 B=32 EMBED_FUSE=$f"; LRAM_EMBED_FUSE=$f run --config xlstm_c1 --batch 32 --steps 400 --warmup 40; done
for f in 1 0; do echo "== headline EMBED_FUSE=$f"; LRAM_EMBED_FUSE=$f run --steps 40 --warmup 8; done
