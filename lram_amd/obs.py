"""Observation layouts the hot path sees and index tables for the device-side front end.

DMControl environments are evaluated with dict observations scattered into a common 204-dim space
(`DmcFullObsWrapper`, src/envs/dmcontrol_utils.py:80-99); the start offsets below are the reference's
`DMC_OBSTYPE_TO_STARTIDX` / `DMC_OBSTYPE_TO_DIM` tables (src/envs/dmcontrol_utils.py:35-49).  Mimicgen low-dim
observations are scattered the same way into a 168-dim space (`MimicgenGymWrapper(to_full_space=True)`,
src/envs/mimicgen_utils.py:58-78,190-214) and then zero-padded to 204; Meta-World and Composuite state
observations are only zero-padded (src/algos/decision_xlstm.py:16-19).  Both tables are pinned against the
reference's own mapping function by tests/golden/reference_vectors.json (`obs_full_space`).
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


def _start_index(table: Dict[str, int]) -> Dict[str, int]:
    out, cum = {}, 0
    for k, v in table.items():
        out[k] = cum
        cum += v
    return out


def dmc_start_index() -> Dict[str, int]:
    return _start_index(DMC_OBSTYPE_TO_DIM)


def inverse_index(obs_spec: Sequence, table: Dict[str, int], state_dim: int) -> torch.Tensor:
    """obs_spec: ordered (key, dim) pairs of the env's flattened dict observation (e.g. cheetah-run:
    [("position", 8), ("velocity", 9)]); table: the domain's key -> slot width table.  Returns int32[state_dim]:
    source column per output dim, -1 = zero (the index table lram_pad_obs consumes)."""
    start = _start_index(table)
    if sum(table.values()) > state_dim:
        raise ValueError("the domain's full observation space is wider than state_dim")
    inv = torch.full((state_dim,), -1, dtype=torch.int32)
    col = 0
    for key, dim in obs_spec:
        if dim > table[key]:
            raise ValueError(f"{key}: {dim} dims exceed the full-space slot of {table[key]}")
        inv[start[key]: start[key] + dim] = torch.arange(col, col + dim, dtype=torch.int32)
        col += dim
    return inv


def dmc_inverse_index(obs_spec: Sequence, state_dim: int = DMC_FULL_OBS_DIM) -> torch.Tensor:
    return inverse_index(obs_spec, DMC_OBSTYPE_TO_DIM, state_dim)


# Mimicgen low-dim observation slots (src/envs/mimicgen_utils.py:58-78): 168 dims, zero-padded to max_state_dim
MIMICGEN_OBSTYPE_TO_DIM: Dict[str, int] = {
    "object": 86, "robot0_eef_pos": 3, "robot0_eef_pos_rel_pod": 3, "robot0_eef_pos_rel_pod_holder": 3,
    "robot0_eef_quat": 4, "robot0_eef_quat_rel_pod": 4, "robot0_eef_quat_rel_pod_holder": 4, "robot0_eef_vel_ang": 3,
    "robot0_eef_vel_lin": 3, "robot0_gripper_qpos": 2, "robot0_gripper_qvel": 2, "robot0_joint_pos": 7,
    "robot0_joint_pos_cos": 7, "robot0_joint_pos_sin": 7, "robot0_joint_vel": 7, "robot0_contact": 1,
    "robot0_eef_force_norm": 1, "robot0_eef_pos_rel_base": 3, "robot0_eef_pos_rel_piece_1": 3,
    "robot0_eef_pos_rel_piece_2": 3, "robot0_eef_quat_rel_base": 4, "robot0_eef_quat_rel_piece_1": 4,
    "robot0_eef_quat_rel_piece_2": 4,
}
MIMICGEN_FULL_OBS_DIM = sum(MIMICGEN_OBSTYPE_TO_DIM.values())  # 168
# MAIN_LOWDIM_KEYS order of the wrapper (mimicgen_utils.py:80); the "object" slot holds up to 86 dims
MIMICGEN_MAIN_LOWDIM_KEYS = ("robot0_eef_pos", "robot0_eef_quat", "robot0_gripper_qpos", "object")


def mimicgen_inverse_index(obs_spec: Sequence, state_dim: int = DMC_FULL_OBS_DIM) -> torch.Tensor:
    return inverse_index(obs_spec, MIMICGEN_OBSTYPE_TO_DIM, state_dim)


def apply_inverse_index(x: torch.Tensor, inv: torch.Tensor) -> torch.Tensor:
    """Host-side statement of what lram_pad_obs does with an index table: out[:, j] = x[:, inv[j]] or 0."""
    inv = inv.to(torch.long)
    out = x[..., inv.clamp(min=0)]
    return torch.where(inv >= 0, out, torch.zeros((), dtype=x.dtype))


CHEETAH_RUN_SPEC = (("position", 8), ("velocity", 9))
