"""lram_amd -- MI355X-native recurrent action-inference engine for LRAM's per-timestep rollout loop.

Only what the hot path needs: `csrc/` (hand-written HIP kernels for gfx950 + the C ABI of
include/lram_hip.h), the ctypes host binding (`engine`), the reference-compatible agent surface (`agent`),
config / checkpoint handling (`config`, `weights`), the batched rollout driver (`rollout`) and the
env-sharded multi-GPU helpers (`dist`).
"""
from .config import ModelSpec, load_agent_params, preset, spec_from_agent_params  # noqa: F401
from .weights import engine_layout, init_state_dict, load_sb3_zip, reference_layout  # noqa: F401

__all__ = ["ModelSpec", "load_agent_params", "preset", "spec_from_agent_params", "engine_layout", "init_state_dict",
           "load_sb3_zip", "reference_layout"]
