#!/bin/bash
for r in 0 64 128 0 64; do echo "== LRAM_BATCHED_F32_ROWS=$r"; LRAM_BATCHED_F32_ROWS=$r PREFILL_MODES=chunkwise python scripts/bench_prefill.py xlstm_206m 64 512 | tail -1; done
for r in 0 64 128; do echo "== 16M b32 LRAM_BATCHED_F32_ROWS=$r"; LRAM_BATCHED_F32_ROWS=$r python bench.py --batch 32 --steps 200 --warmup 20 --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; done
