"""Prefill micro-benchmark: L stored timesteps through lram_prefill (chunks of 4 timesteps per state pass) vs the
same L timesteps as L lram_step calls.  Prints timesteps/s per env batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine
cfg, B, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spec = preset(cfg); sd = init_state_dict(spec, 0)
eng = Engine(spec, sd, B, device="cuda:0")
obs = torch.rand(B, L, spec.state_dim, device="cuda:0") * 2 - 1
rtg = torch.full((B, L), 4.5, device="cuda:0"); rew = torch.zeros(B, L, device="cuda:0")
obs_t = [obs[:, l].contiguous() for l in range(L)]; rtg_t = [rtg[:, l].contiguous() for l in range(L)]; rew_t = [rew[:, l].contiguous() for l in range(L)]
def seq():
    for l in range(L): eng.step(obs_t[l], rtg_t[l], rew_t[l], None)
def pre():
    eng.prefill(obs, rtg, rew)
for name, fn in (("sequential steps", seq), ("lram_prefill", pre), ("sequential steps", seq), ("lram_prefill", pre)):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{cfg} B={B} L={L} {name}: {dt*1e3:.1f} ms  {B*L/dt:,.0f} env-timesteps/s", flush=True)
