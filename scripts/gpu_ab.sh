#!/bin/bash
# A/B runs of the headline bench on ONE box: each line "VAR=val ..." of the arguments is one configuration.
# usage: bash scripts/gpu_ab.sh "LRAM_LAZY_NARROW=0" "LRAM_LAZY_NARROW=1" ...
mkdir -p gpurun_out
out=gpurun_out/ab.txt
: > $out
for rep in 1 2; do
for cfg in "$@"; do
  line=$(env $cfg timeout 300 python bench.py --no-cpu-baseline --no-stream-ceilings --steps 48 --warmup 8 ${BENCH_ARGS} 2>/dev/null | tail -1)
  echo "$cfg | $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline",{}); print(round(d["value"]), d["ms_per_step"], round(r.get("state_pass_avg_ms",0),4), round(r.get("fold_avg_ms",0),4))' 2>/dev/null)" | tee -a $out
done
done
