#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "slstm_token or xlstm_16m_shapes or two_blocks" 2>&1 | tail -5
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 12 32 64 128 256 512; do for r in 512 0; do echo "== 16M B=$b LRAM_SLSTM_FUSED_ROWS=$r"; LRAM_SLSTM_FUSED_ROWS=$r run --batch $b --steps 150 --warmup 20 --micro 1; done; done
for b in 16 64 256; do for r in 512 0; do echo "== 206M B=$b LRAM_SLSTM_FUSED_ROWS=$r"; LRAM_SLSTM_FUSED_ROWS=$r run --config xlstm_206m --batch $b --steps 40 --warmup 5 --micro 1; done; done
for r in 256 0; do echo "== prefill 206M LRAM_SLSTM_FUSED_ROWS=$r"; LRAM_SLSTM_FUSED_ROWS=$r PREFILL_MODES=chunkwise python scripts/bench_prefill.py xlstm_206m 64 512 | tail -1; done
for r in 256 0; do echo "== prefill 16M 128x63 LRAM_SLSTM_FUSED_ROWS=$r"; LRAM_SLSTM_FUSED_ROWS=$r PREFILL_MODES=chunkwise python scripts/bench_prefill.py xlstm_16m 128 63 | tail -1; done
