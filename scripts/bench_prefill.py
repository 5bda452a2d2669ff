"""Prefill micro-benchmark: L stored timesteps through lram_prefill -- chunkwise matrix-core kernels (up to 21
timesteps per state pass) and token-sequential kernels (4 per pass, LRAM_PREFILL_CHUNK=0) -- vs the same L
timesteps as L lram_step calls.  Prints env-timesteps/s per mode.

    python scripts/bench_prefill.py <config> <B> <L> [micro]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine

cfg, B, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
micro = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # env slices (0 = auto)
spec = preset(cfg); sd = init_state_dict(spec, 0)
dev = "cuda:0"
eng = Engine(spec, sd, B, device=dev)
os.environ["LRAM_PREFILL_CHUNK"] = "0"
eng_seq = Engine(spec, sd, B, device=dev)
del os.environ["LRAM_PREFILL_CHUNK"]
for e in (eng, eng_seq):
    e.set_micro_batches(micro)
obs = torch.rand(B, L, spec.state_dim, device=dev) * 2 - 1
rtg = torch.full((B, L), 4.5, device=dev); rew = torch.zeros(B, L, device=dev)
obs_t = [obs[:, l].contiguous() for l in range(L)]; rtg_t = [rtg[:, l].contiguous() for l in range(L)]
rew_t = [rew[:, l].contiguous() for l in range(L)]


def steps():
    for l in range(L):
        eng.step(obs_t[l], rtg_t[l], rew_t[l], None)


modes = (("lram_step x L", steps), ("lram_prefill, token-sequential chunks of 4", lambda: eng_seq.prefill(obs, rtg, rew)),
         ("lram_prefill, chunkwise", lambda: eng.prefill(obs, rtg, rew)))
only = os.environ.get("PREFILL_MODES")   # e.g. "chunkwise": restrict to modes whose name contains the string
if only:
    modes = tuple(m for m in modes if only in m[0])
for rep in range(2):
    for name, fn in modes:
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep == 1:
            print(f"{cfg} B={B} L={L} {name}: {dt*1e3:.1f} ms  {B*L/dt:,.0f} env-timesteps/s", flush=True)
