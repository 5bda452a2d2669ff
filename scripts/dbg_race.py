import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine
from tests.test_gpu_fullsize import _inputs
spec = preset("xlstm_16m"); sd = init_state_dict(spec, 0)
B = 4096
seq = _inputs(spec, B, 6, seed=2024)
dseq = [[t.cuda() for t in x] for x in seq]
def run(micro):
    eng = Engine(spec, sd, B, device="cuda:0"); eng.set_micro_batches(micro)
    out = []
    for x in dseq:
        a, _ = eng.step(*x); torch.cuda.synchronize(); out.append((a.clone(), eng.taps()[1].clone()))
    st = eng.export_state_tensor(0, 0).clone(); eng.close(); torch.cuda.empty_cache()
    return out, st
ref, cref = run(1)
for rep in range(3):
    got, c = run(2)
    bad = [(t, int((g[0] != r[0]).sum()), float((g[1]-r[1]).abs().max())) for t, (g, r) in enumerate(zip(got, ref))]
    print("rep", rep, "mismatching actions / max hidden diff per step:", bad, "C diff", float((c-cref).abs().max()), flush=True)
print("---- locate")
for rep in range(3):
    eng = Engine(spec, sd, B, device="cuda:0"); eng.set_micro_batches(2)
    a, _ = eng.step(*dseq[0]); torch.cuda.synchronize()
    tok, hid, lg = eng.taps()
    d = (hid - ref[0][1]).abs().amax(dim=(1, 2))
    bad = (d > 1e-3).nonzero().flatten().tolist()
    print("rep", rep, "n bad envs", len(bad), "first", bad[:24], "tok diff", float((tok - tok).abs().max()))
    for blk in (0, 2, 3):
        pass
    eng.close(); torch.cuda.empty_cache()
