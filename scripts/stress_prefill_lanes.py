"""Race screen of lram_prefill's chunk lanes: prefills with three chunks in flight against one chunk at a time (LRAM_PREFILL_CHUNK=3) on
fresh random contexts -- actions and exported states must be BIT-identical.  python scripts/stress_prefill_lanes.py (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine
for cfg, B, L, reps in (("xlstm_206m", 64, 512, 6), ("xlstm_16m", 256, 200, 10), ("xlstm_206m", 200, 130, 4), ("mamba_48m", 256, 120, 8)):
    spec = preset(cfg); sd = init_state_dict(spec, 0)
    os.environ["LRAM_PREFILL_CHUNK"] = "3"; e_ser = Engine(spec, sd, B, device="cuda:0")
    os.environ["LRAM_PREFILL_CHUNK"] = "1"; e_lan = Engine(spec, sd, B, device="cuda:0")
    g = torch.Generator(device="cuda").manual_seed(1)
    bad = 0
    for r in range(reps):
        obs = torch.rand(B, L, spec.state_dim, device="cuda", generator=g) * 2 - 1
        rtg = torch.full((B, L), 4.5, device="cuda"); rew = torch.zeros(B, L, device="cuda")
        mask = torch.ones(B, dtype=torch.uint8, device="cuda") if r % 3 == 0 else None
        a1, _ = e_ser.prefill(obs, rtg, rew, mask); a1 = a1.clone()
        a2, _ = e_lan.prefill(obs, rtg, rew, mask); a2 = a2.clone()
        torch.cuda.synchronize()
        same = torch.equal(a1, a2)
        for blk in (0, spec.n_blocks - 1):
            for which in ((0, 3) if (spec.backbone == "mamba" or blk in spec.slstm_at) else (0, 1, 2, 3)):
                same = same and torch.equal(e_ser.export_state_tensor(blk, which), e_lan.export_state_tensor(blk, which))
        bad += 0 if same else 1
    print(cfg, B, L, "reps", reps, "mismatching reps", bad, flush=True)
    e_ser.close(); e_lan.close(); torch.cuda.empty_cache()
