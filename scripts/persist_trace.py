"""Per-phase timing of the whole-step kernel (workgroup 0's view): work time and barrier wait per phase."""
import os, sys
os.environ["LRAM_PERSIST_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine

cfg, B = sys.argv[1], int(sys.argv[2])
spec = preset(cfg); sd = init_state_dict(spec, 0)
eng = Engine(spec, sd, B, device="cuda:0")
obs = torch.rand(B, spec.state_dim, device="cuda") * 2 - 1
rtg, rew = torch.full((B,), 4.0, device="cuda"), torch.zeros(B, device="cuda")
for _ in range(20):
    eng.step(obs, rtg, rew, None)
n_m = spec.n_blocks - len(spec.slstm_at); n_s = len(spec.slstm_at)
nb = 2 + 4 * n_m + 6 * n_s
t = eng.persistent_trace(2 * nb + 1)
names = ["front"]
for i in range(spec.n_blocks):
    names += [f"b{i}.S{k}" for k in range(1, 7)] if i in spec.slstm_at else [f"b{i}.{k}" for k in "ABCD"]
names += ["head"]
work = [(t[2 * i + 1] - t[2 * i]) / 100.0 for i in range(nb)]
wait = [(t[2 * i + 2] - t[2 * i + 1]) / 100.0 for i in range(nb)]
print(f"{cfg} B={B} wgs={os.environ.get('LRAM_PERSIST_WGS', '128')}: total {(t[2 * nb] - t[0]) / 100.0:.1f} us; work {sum(work):.1f} us, barrier wait {sum(wait):.1f} us")
for n, w, b in list(zip(names, work, wait))[:12] + list(zip(names, work, wait))[-3:]:
    print(f"  {n:8s} work {w:7.2f} us   barrier {b:7.2f} us")
