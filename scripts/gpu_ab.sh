#!/bin/bash
# A/B runs of the headline bench on ONE box, two rounds.  Each argument is one configuration:
#   "VAR=val VAR2=val2 -- --bench-flag x"   (environment before " -- ", extra bench.py flags after it; either may be empty)
# usage: bash scripts/gpu_ab.sh "LRAM_GN_FUSE=0" "LRAM_GN_FUSE=1 -- --micro 3"
mkdir -p gpurun_out
out=gpurun_out/ab.txt
: > $out
for rep in 1 2; do
for cfg in "$@"; do
  envp="${cfg%% -- *}"; argp=""
  case "$cfg" in *" -- "*) argp="${cfg#* -- }";; esac
  case "$cfg" in "-- "*) envp=""; argp="${cfg#-- }";; esac
  line=$(env $envp timeout 300 python bench.py --no-cpu-baseline --no-stream-ceilings --steps 48 --warmup 8 ${BENCH_ARGS} $argp 2>/dev/null | tail -1)
  echo "$cfg | $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline",{}); print(round(d["value"]), round(d["ms_per_step"],3), round(r.get("state_pass_avg_ms",0),4), round(r.get("fold_avg_ms",0),4))' 2>/dev/null)" | tee -a $out
done
done
