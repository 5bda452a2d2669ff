"""Where does the hidden-state error of a weight distribution come from?  Sub-stacks of the first n blocks over a trajectory:
engine vs float64 oracle beside fp32 oracle vs float64 oracle, per step (worst row), plus the state error at the end.

    python scripts/debug_scheme.py <preset> <scheme> <B> <steps> [n ...]      (LRAM_* knobs apply)
"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine
from oracle.dt_ref import OraclePolicy
from tests.helpers import Fp64Oracle, make_inputs

name, scheme, B, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
spec = preset(name)
sd = init_state_dict(spec, seed=0, scheme=scheme)
seq = make_inputs(spec, B, steps, seed=1234, reset_prob=0.03)
if scheme == "trained_like":
    for s in seq:
        s[0][:, 3] *= 30.0
for n in [int(x) for x in (sys.argv[5:] or range(1, spec.n_blocks + 1))]:
    sp = dataclasses.replace(spec, n_blocks=n, slstm_at=[i for i in spec.slstm_at if i < n])
    eng = Engine(sp, sd, B, device="cuda:0")
    ora, o64 = OraclePolicy(sp, sd), Fp64Oracle(sp, sd)
    worst = (0.0, 0.0, -1)
    line = []
    for t, (obs, rtg, rew, mask) in enumerate(seq):
        eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        _, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        _, d64 = o64.step(obs, rtg, rew, mask, return_debug=True)
        torch.cuda.synchronize()
        _, hid, _ = eng.taps()
        sc = float(d64["hidden"].abs().max())
        e64 = float((hid.cpu().double() - d64["hidden"]).abs().max()) / sc
        o32 = float((dbg["hidden"].double() - d64["hidden"]).abs().max()) / sc
        line.append(f"{e64:.0e}/{o32:.0e}")
        if e64 > worst[0]:
            worst = (e64, o32, t)
    print(f"n={n:2d} worst step {worst[2]}: engine vs fp64 {worst[0]:.2e}, fp32 oracle vs fp64 {worst[1]:.2e} | " + " ".join(line[::3]), flush=True)
    i = n - 1
    if i not in sp.slstm_at:
        for j, nm in enumerate("Cnm"):
            got = eng.export_state_tensor(i, j).cpu().double().flatten()
            w64 = o64.ora.state[f"block_{i}"]["mlstm_state"][j].double().flatten()
            w32 = ora.state[f"block_{i}"]["mlstm_state"][j].double().flatten()
            sc = float(w64.abs().max()) + 1e-30
            print(f"      block {i} {nm}: engine vs fp64 {float((got - w64).abs().max()) / sc:.2e}, oracle vs fp64 {float((w32 - w64).abs().max()) / sc:.2e}, "
                  f"range [{float(w64.min()):.3g}, {float(w64.max()):.3g}]")
    eng.close()
