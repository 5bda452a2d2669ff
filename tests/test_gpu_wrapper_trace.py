"""GPU: the reference's xLSTM inference wrapper, executed (tests/golden/make_golden_from_reference.py::xlstm_model_trace:
MultiDomainDiscreteDecisionXLSTMModel.forward / compute_hidden_states / handle_inference_cache / xLSTMEncoder.forward /
get_predictions under the executed agent chain predict -> pad_inputs -> get_action_pred), replayed through
RecurrentAgent's reference surface on the HIP engine: same actions (continuous heads: within 1e-4 = same bin) over
reset_inf_cache_freq cache drops and an episode end."""
import dataclasses
import json
import os

import pytest
import torch

from tests.test_oracle_golden import _xlstm_trace_env, run_agent_over_xlstm_trace

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_agent_on_the_engine_reproduces_the_executed_xlstm_wrapper(hip_lib):
    from lram_amd.agent import RecurrentAgent
    with open(os.path.join(GOLD, "reference_vectors.json")) as fh:
        v = json.load(fh)["xlstm_model_trace"]
    agents = []

    def make(spec, sd):
        spec = dataclasses.replace(spec, reset_inf_cache_freq=v["reset_inf_cache_freq"])
        agents.append(RecurrentAgent(spec, sd, n_envs=1, device="cuda:0"))
        return agents[-1]

    for e in range(len(v["envs"])):
        assert run_agent_over_xlstm_trace(make, v, e, atol=1e-4) <= 1e-4
        # hidden state of the last step against the executed reference's last_encoder_output
        env = v["envs"][e]
        _, hidden, _ = agents[-1].engine.taps()
        ref = torch.tensor(env["hidden"][-1])
        assert float((hidden[0].cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), e
        agents[-1].engine.close()
