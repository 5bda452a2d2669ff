"""Host-side cost of enqueueing one env-step (C-ABI call returning, nothing synchronised) vs the device time per step."""
import sys, time
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = preset(sys.argv[2] if len(sys.argv) > 2 else "xlstm_16m")
eng = Engine(spec, init_state_dict(spec, seed=0), B, device="cuda:0")
obs = torch.rand(B, spec.state_dim, device="cuda") * 2 - 1
rtg = torch.full((B,), 4.5, device="cuda")
rew = torch.zeros(B, device="cuda")
mask = torch.zeros(B, dtype=torch.uint8, device="cuda")
for _ in range(30):
    eng.step(obs, rtg, rew, mask)
torch.cuda.synchronize()
n = 64
t0 = time.perf_counter()
per = []
for _ in range(n):
    a = time.perf_counter()
    eng.step(obs, rtg, rew, mask)
    per.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
per.sort()
print(f"B={B}: enqueue {1e3 * (t1 - t0) / n:.3f} ms/step (median call {1e3 * per[n // 2]:.3f}, min {1e3 * per[0]:.3f}), "
      f"enqueue + drain {1e3 * (t2 - t0) / n:.3f} ms/step")
# with a host synchronisation after every step: enqueue time is exposed only where the device is faster than the host
t0 = time.perf_counter()
for _ in range(n):
    eng.step(obs, rtg, rew, mask)
    torch.cuda.current_stream().synchronize()
t1 = time.perf_counter()
print(f"B={B}: host-synchronised {1e3 * (t1 - t0) / n:.3f} ms/step")
