#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -m gpu -x -k "lazy or full or pipeline or slice" 2>&1 | tail -8
B="python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-stream-ceilings"
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]
print(f'{sys.argv[1].split("/")[-1]:28s} value {d["value"]:9.0f}  host_io {d.get("host_io",{}).get("value",0):9.0f}  pass {r.get("state_pass_avg_ms",0):.3f} fold {r.get("fold_avg_ms",0):.3f}  standalone pass {r.get("standalone",{}).get("state_pass_avg_ms",0):.3f}')
PY
}
$B > $OUT/b_default.json 2>/dev/null; show $OUT/b_default.json
LRAM_LAZY_UNROLL=8 $B > $OUT/b_unr8.json 2>/dev/null; show $OUT/b_unr8.json
LRAM_LAZY_UNROLL=4 $B > $OUT/b_unr4.json 2>/dev/null; show $OUT/b_unr4.json
LRAM_LAZY_KPREFETCH=0 $B > $OUT/b_nopre.json 2>/dev/null; show $OUT/b_nopre.json
GPU_MAX_HW_QUEUES=8 $B > $OUT/b_q8.json 2>/dev/null; show $OUT/b_q8.json
