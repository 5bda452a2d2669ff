#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for f in 1 0 1 0; do echo "== 16M B=1 graph EMBED_FUSE=$f"; LRAM_EMBED_FUSE=$f run --batch 1 --steps 400 --warmup 40 --graph; done
for f in 1 0; do echo "== 16M B=1 eager EMBED_FUSE=$f"; LRAM_EMBED_FUSE=$f run --batch 1 --steps 400 --warmup 40; done
for f in 1 0; do echo "== mamba B=1 graph EMBED_FUSE=$f"; LRAM_EMBED_FUSE=$f run --config mamba_48m --batch 1 --steps 200 --warmup 20 --graph; done
timeout 600 python -m pytest tests/test_gpu_lazy.py -q -m gpu -x 2>&1 | tail -2
