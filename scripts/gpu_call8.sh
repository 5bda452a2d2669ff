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
for S in 0 1 0 1; do LRAM_SPLIT_UP=$S $B > $OUT/b_split$S.json 2>/dev/null; show $OUT/b_split$S.json; done
LRAM_SPLIT_UP=0 $B --state eager > $OUT/b_eager_split0.json 2>/dev/null; show $OUT/b_eager_split0.json
LRAM_SPLIT_UP=1 $B --state eager > $OUT/b_eager_split1.json 2>/dev/null; show $OUT/b_eager_split1.json
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lazy.py tests/test_gpu_fullsize.py tests/test_gpu_chunk.py tests/test_gpu_edge.py -q -m gpu -x 2>&1 | tail -3
