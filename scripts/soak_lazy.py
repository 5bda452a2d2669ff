"""Long-run agreement of the lazy and the materialised matrix memory: N steps with random resets, same inputs;
reports action mismatches and the relative distance of the exported states at the end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine

cfg, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spec = preset(cfg); sd = init_state_dict(spec, 3)
dev = "cuda:0"
eng = {m: Engine(spec, sd, B, device=dev) for m in ("eager", "lazy")}
for m, e in eng.items():
    e.set_state_mode(m)
g = torch.Generator(device=dev).manual_seed(7)
mism, worst = 0, 0.0
rtg = torch.full((B,), 4.5, device=dev)
for t in range(N):
    obs = torch.rand(B, spec.state_dim, generator=g, device=dev) * 2 - 1
    mask = (torch.rand(B, generator=g, device=dev) < (1.0 if t == 0 else 0.01)).to(torch.uint8)
    rtg = torch.where(mask.bool(), torch.full_like(rtg, 4.5), rtg - 0.01)
    rew = torch.zeros(B, device=dev)
    a = {m: e.step(obs, rtg, rew, mask)[0].clone() for m, e in eng.items()}
    d = (a["eager"] - a["lazy"]).abs()
    mism += int((d > 1e-4).sum())
torch.cuda.synchronize()
for blk in range(spec.n_blocks):
    if blk in spec.slstm_at:
        continue
    for which in (0, 1, 2):
        x, y = eng["eager"].export_state_tensor(blk, which), eng["lazy"].export_state_tensor(blk, which)
        worst = max(worst, float((x - y).abs().max() / (x.abs().max() + 1e-12)))
print(f"{cfg} B={B} N={N}: action elements differing by more than 1e-4: {mism} of {N * B * spec.act_dim}; "
      f"worst relative state distance {worst:.2e}")
