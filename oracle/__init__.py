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
* Reference-repo files on the path that can be imported in the build container
  (``src/tokenizers_custom/minmax_tokenizer.py``, ``src/algos/models/rms_norm.py``)
  were executed there and their input/output vectors are committed under
  ``tests/golden/`` (``make_golden_from_reference.py`` is the generating script).
  ``oracle.dt_ref`` is pinned against them.
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
