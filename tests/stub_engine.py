"""Stand-in for lram_amd.engine.Engine inside bench.main on CPU (test infrastructure; bench.py --engine-factory)."""
import os


class StubEngine:
    """Stands in for lram_amd.engine.Engine inside bench.main on CPU: actions are a fixed function of the inputs, so
    the sharded + gathered result can be compared with a single-process run."""

    def __init__(self, spec, batch, device):
        self.spec, self.batch, self.device = spec, batch, device
        self.state_mode = "materialised"
        self.steps = 0

    def set_micro_batches(self, n):
        pass

    def set_graph_mode(self, on):
        pass

    def step(self, obs, rtg, reward, reset_mask=None, **kw):
        self.steps += 1
        a = obs[:, :self.spec.act_dim] * 0.5 + rtg.view(-1, 1) + reset_mask.float().view(-1, 1)
        return a, None


def factory(spec, batch, device):
    return StubEngine(spec, batch, device)


def factory_rank1_dies(spec, batch, device):
    if os.environ.get("RANK") == "1":
        raise SystemExit(7)
    return StubEngine(spec, batch, device)
