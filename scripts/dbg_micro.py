import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict, preset
from lram_amd.engine import Engine
from tests.helpers import make_inputs
spec = preset("xlstm_tiny"); sd = init_state_dict(spec, 13); B = 11
seq = make_inputs(spec, B, 3, seed=99)
for n, graph in ((1, False), (2, False), (3, False), (2, True), (4, True)):
    print("config", n, graph, flush=True)
    eng = Engine(spec, sd, B, device="cuda:0")
    eng.set_micro_batches(n); eng.set_graph_mode(graph)
    d = [torch.empty_like(t).cuda() for t in seq[0]]
    for i, inp in enumerate(seq):
        for dst, src in zip(d, inp): dst.copy_(src)
        a, _ = eng.step(*d); torch.cuda.synchronize()
        print("  step", i, float(a.sum()), flush=True)
    x = torch.randn(B, 2, spec.d_model).cuda()
    enc = eng.encoder_step(x); torch.cuda.synchronize(); print("  enc ok", flush=True)
    eng.close(); print("  closed", flush=True)
