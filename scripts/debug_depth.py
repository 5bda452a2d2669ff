"""Where does the 206M stack's hidden-state error against the oracle come from?  Sub-stacks of the first n blocks."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine
from oracle.dt_ref import OraclePolicy
from tests.helpers import Fp64Oracle, make_inputs

spec = preset("xlstm_206m")
sd = init_state_dict(spec, seed=0, with_image_encoder=True)
sd = {k: v for k, v in sd.items() if not k.startswith("embed_image.")}
B = 3
seq = make_inputs(spec, B, 2, seed=1234)
for n in [int(x) for x in (sys.argv[1:] or [1, 2, 3, 4, 6, 8, 12, 16, 20])]:
    sp = dataclasses.replace(spec, n_blocks=n, slstm_at=[i for i in spec.slstm_at if i < n])
    eng = Engine(sp, sd, B, device="cuda:0")
    ora = OraclePolicy(sp, sd)
    o64 = Fp64Oracle(sp, sd)
    for t, (obs, rtg, rew, mask) in enumerate(seq):
        eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        _, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        _, d64 = o64.step(obs, rtg, rew, mask, return_debug=True)
        torch.cuda.synchronize()
        _, hid, _ = eng.taps()
        e = (hid.cpu() - dbg["hidden"]).abs()
        per = e.amax(dim=-1)  # [B, T]
        print(f"n={n:2d} step {t}: max abs err {float(e.max()):.3e} (ref max {float(dbg['hidden'].abs().max()):.2f}) per (env, token): "
              + " ".join(f"{x:.1e}" for x in per.flatten().tolist()), flush=True)
        e64 = (hid.cpu().double() - d64["hidden"]).abs().amax(dim=-1)
        o32 = (dbg["hidden"].double() - d64["hidden"]).abs().amax(dim=-1)
        print("      engine vs fp64: " + " ".join(f"{x:.1e}" for x in e64.flatten().tolist()))
        print("      oracle vs fp64: " + " ".join(f"{x:.1e}" for x in o32.flatten().tolist()), flush=True)
    eng.close()
