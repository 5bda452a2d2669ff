"""CPU oracle for LRAM's per-timestep action-inference path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

This package is a plain PyTorch-eager fp32 restatement (CPU) of the arithmetic on
the reference's rollout hot path:

  evaluate.py -> custom_evaluate_policy -> agent.predict -> policy.forward(
      use_inference_cache=True) -> xLSTMBlockStack.step / Mamba.step -> action head

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  ``lram_amd`` (the product)
never imports it; the product path raises when the HIP library is missing.

PARITY PINNING STATUS
---------------------
* Reference code on the path that can be executed in the build container was executed
  there and its input/output vectors are committed under ``tests/golden/``
  (``make_golden_from_reference.py`` is the generating script): the action tokenizer and
  ``LlamaRMSNorm`` (importable files), and -- lifted out of modules whose imports need
  gym / stable_baselines3 by executing only the named classes / methods -- the token front
  end (``compute_inputs`` ... ``prepare_inputs_and_masks``), the head post-processing
  (``prepare_action_logits`` / ``get_action_from_logits``), the ImpalaCNN block modules and
  the DMControl / Mimicgen full-space observation mapping.  ``oracle.dt_ref`` is pinned
  against them (tests/test_oracle_golden.py).  Round 2 added the Mamba agent's control flow
  (``DiscreteDecisionMamba.get_action_pred`` + ``InferenceParams.reset`` + ``MambaEncoder.forward``
  executed over two envs x two episodes; ``OraclePolicy(mamba_repeat=..., stale_state=True)``
  reproduces the returned actions exactly) and the checkpoint key handling of
  ``load_model_weights`` (checked by tests/test_config_weights.py against lram_amd.weights).
* The recurrent arithmetic itself lives in third-party, un-vendored packages that
  are NOT under /root/reference and NOT installed here: ``xlstm`` (unpinned,
  reference README.md:94-97, API of 1.0.x), ``mamba_ssm==2.1.0``,
  ``causal-conv1d==1.3.0.post1`` (README.md:99-103).  ``oracle.xlstm_ref`` and
  ``oracle.mamba_ref`` restate their published algorithms.  The reference holds no
  tests / golden vectors for that boundary  =>  **parity unpinned** for the xLSTM
  backbone.  What backs it instead: step<->parallel-form equivalence, batched<->
  single-env equivalence, and (Mamba only) agreement with the independent
  ``transformers.models.mamba`` pure-torch implementation that is installed here.
"""
