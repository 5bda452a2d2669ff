import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine
from oracle.dt_ref import OraclePolicy
from tests.test_gpu_fullsize import _inputs
spec = preset("xlstm_16m"); sd = init_state_dict(spec, 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
micro = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seq = _inputs(spec, 4096, 4, seed=2024)
idx = torch.arange(0, 4)
eng = Engine(spec, sd, B, device="cuda:0"); eng.set_micro_batches(micro)
ora = OraclePolicy(spec, sd)
for t, (obs, rtg, rew, mask) in enumerate(seq):
    a, _ = eng.step(obs[:B].cuda(), rtg[:B].cuda(), rew[:B].cuda(), mask[:B].cuda()); torch.cuda.synchronize()
    ref, dbg = ora.step(obs[idx], rtg[idx], rew[idx], mask[idx], return_debug=True)
    tok, hid, lg = eng.taps()
    errs = []
    for blk in (0, 2, 7):
        c = eng.export_state_tensor(blk, 0)[:4].cpu(); cr = ora.state[f"block_{blk}"]["mlstm_state"][0]
        n = eng.export_state_tensor(blk, 1)[:4].cpu(); nr = ora.state[f"block_{blk}"]["mlstm_state"][1]
        m = eng.export_state_tensor(blk, 2)[:4].cpu(); mr = ora.state[f"block_{blk}"]["mlstm_state"][2]
        errs.append((blk, float((c-cr).abs().max()/cr.abs().max()), float((n-nr).abs().max()/nr.abs().max()), float((m-mr).abs().max())))
    print(t, "mask", mask[idx].tolist(), "hid", float((hid[:4].cpu()-dbg["hidden"]).abs().max()/dbg["hidden"].abs().max()), errs, flush=True)
