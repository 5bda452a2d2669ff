"""Experiment: do two half-batch engines on two streams overlap (HBM-bound cell kernel of one half beside
the MFMA-bound GEMMs of the other)?  Prints env-steps/s for 1x4096 vs 2x2048 (two streams) vs 4x1024."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine

spec = preset("xlstm_16m")
sd = init_state_dict(spec, 0)
dev = torch.device("cuda:0")
TOTAL = 4096

def run(nsplit, steps=24, warm=4):
    B = TOTAL // nsplit
    engs = [Engine(spec, sd, B, device=dev) for _ in range(nsplit)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
    obs = [torch.rand(B, spec.state_dim, device=dev) * 2 - 1 for _ in range(nsplit)]
    rtg = [torch.full((B,), 4.5, device=dev) for _ in range(nsplit)]
    rew = [torch.zeros(B, device=dev) for _ in range(nsplit)]
    torch.cuda.synchronize()
    def step():
        for i in range(nsplit):
            with torch.cuda.stream(streams[i]):
                engs[i].step(obs[i], rtg[i], rew[i], None)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in engs:
        e.close()
    torch.cuda.empty_cache()
    return TOTAL * steps / dt, dt / steps * 1e3

for n in (1, 4, 2, 4, 8, 4, 8, 2, 1):
    v, ms = run(n)
    print(f"nsplit={n}: {v:,.0f} env-steps/s  {ms:.2f} ms/step", flush=True)
