#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-stream-ceilings --host-io-steps 0"
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]
print(f'{sys.argv[1].split("/")[-1]:28s} value {d["value"]:9.0f}  pass {r.get("state_pass_avg_ms",0):.3f} fold {r.get("fold_avg_ms",0):.3f}  standalone pass {r.get("standalone",{}).get("state_pass_avg_ms",0):.3f}')
PY
}
for U in 4 16; do for P in 0 40 56 84; do
  LRAM_LAZY_UNROLL=$U LRAM_CELL_LDS_PAD_KB=$P $B > $OUT/b_u${U}_p${P}.json 2>/dev/null; show $OUT/b_u${U}_p${P}.json
done; done
LRAM_LAZY_UNROLL=4 $B > $OUT/b_u4_again.json 2>/dev/null; show $OUT/b_u4_again.json
