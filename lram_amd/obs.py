"""Observation layouts the hot path sees and index tables for the device-side front end.

DMControl environments are evaluated with dict observations scattered into a common 204-dim space
(`DmcFullObsWrapper`, src/envs/dmcontrol_utils.py:80-99); the start offsets below are the reference's
`DMC_OBSTYPE_TO_STARTIDX` / `DMC_OBSTYPE_TO_DIM` tables (src/envs/dmcontrol_utils.py:35-49).  Meta-World,
Composuite and Mimicgen state observations are zero-padded (src/algos/decision_xlstm.py:16-19).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch

DMC_OBSTYPE_TO_DIM: Dict[str, int] = {
    "orientations": 14, "velocity": 27, "position": 8, "touch": 5, "target_position": 2, "dist_to_target": 1,
    "joint_angles": 21, "upright": 1, "target": 3, "head_height": 1, "extremities": 12, "torso_vertical": 3,
    "com_velocity": 3, "arm_pos": 16, "arm_vel": 8, "hand_pos": 4, "object_pos": 4, "object_vel": 3, "target_pos": 4,
    "orientation": 2, "to_target": 2, "joints": 14, "body_velocities": 45, "height": 1,
}
DMC_FULL_OBS_DIM = sum(DMC_OBSTYPE_TO_DIM.values())  # 204


def dmc_start_index() -> Dict[str, int]:
    out, cum = {}, 0
    for k, v in DMC_OBSTYPE_TO_DIM.items():
        out[k] = cum
        cum += v
    return out


def dmc_inverse_index(obs_spec: Sequence, state_dim: int = DMC_FULL_OBS_DIM) -> torch.Tensor:
    """obs_spec: ordered (key, dim) pairs of the env's flattened dict observation (e.g. cheetah-run:
    [("position", 8), ("velocity", 9)]).  Returns int32[state_dim]: source column per output dim, -1 = zero."""
    start = dmc_start_index()
    inv = torch.full((state_dim,), -1, dtype=torch.int32)
    col = 0
    for key, dim in obs_spec:
        if dim > DMC_OBSTYPE_TO_DIM[key]:
            raise ValueError(f"{key}: {dim} dims exceed the full-space slot of {DMC_OBSTYPE_TO_DIM[key]}")
        inv[start[key]: start[key] + dim] = torch.arange(col, col + dim, dtype=torch.int32)
        col += dim
    return inv


CHEETAH_RUN_SPEC = (("position", 8), ("velocity", 9))
