#!/bin/bash
run() { python bench.py --no-cpu-baseline --host-io-steps 0 --no-stream-ceilings "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
for b in 128 192 256 341; do for r in 1024 0; do echo "== 16M B=$b SKINNY64_ROWS=$r"; LRAM_GEMM_SKINNY64_ROWS=$r run --batch $b --steps 100 --warmup 10; done; done
for b in 64 256; do for r in 1024 0; do echo "== mamba B=$b SKINNY64_ROWS=$r"; LRAM_GEMM_SKINNY64_ROWS=$r run --config mamba_48m --batch $b --steps 100 --warmup 10; done; done
python - <<'PY'
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine
import os
spec = preset("xlstm_16m"); sd = init_state_dict(spec, 0)
B = 200
g = torch.Generator().manual_seed(0)
obs = (torch.rand(B, spec.state_dim, generator=g) * 2 - 1).cuda(); rtg = torch.full((B,), 3.0).cuda(); rew = torch.zeros(B).cuda()
outs = []
for r in ("1024", "0"):
    os.environ["LRAM_GEMM_SKINNY64_ROWS"] = r
    e = Engine(spec, sd, B, device="cuda:0")
    for _ in range(4):
        a, _ = e.step(obs, rtg, rew, None)
    torch.cuda.synchronize(); outs.append(a.clone()); e.close()
print("max action diff 64-row kernel vs default:", float((outs[0] - outs[1]).abs().max()))
PY
