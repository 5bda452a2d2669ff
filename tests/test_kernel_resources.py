"""CPU (hipcc cross-compiles): register / LDS budgets the engine's co-residency reasoning relies on.

Several launch decisions assume how many workgroups of one kernel fit on a CU BESIDE another kernel's (DESIGN.md section 0 item 3,
profiles/r05_ab_206m_chain.txt, r05_ab_read_pass_lds_cap.txt):
  * the sliced-head read pass (206M geometry) is held at two workgroups per CU by its REGISTER count (> 170 VGPRs: twelve registers
    are kept live across the pass for exactly this), so that the other slice's projection workgroups find room;
  * those projection workgroups must fit into what two such read passes leave: the two-stage pre-split GEMM <= 128 VGPRs and about
    65 KB of LDS, the two-stage on-the-fly GEMM <= 160 VGPRs and 48 KB;
  * the 256-wide-head read pass stays at <= 128 VGPRs (three workgroups per CU at 41 KB of LDS, two at 54 KB).
A compiler upgrade that moves these numbers silently changes occupancy, not results -- no parity test would notice."""
import os
import re
import shutil
import subprocess

import pytest

from lram_amd import build

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lram_amd", "csrc")


def _resources(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    cmd = [hipcc] + list(build.FLAGS) + ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src),
                                         "-o", os.devnull]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900).stderr
    res, name = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|LDS Size \[bytes/block\]|VGPRs Spill):\s+(\d+)", line)
        if m and name:
            res[name][m.group(1)] = int(m.group(2))
    assert res, out[-2000:]
    return res


def _one(res, *needles):
    hits = [k for k in res if all(n in k for n in needles)]
    assert len(hits) == 1, (needles, hits)
    return res[hits[0]]


def test_read_pass_register_budgets():
    res = _resources("mlstm_lazy.hip")
    sliced = _one(res, "mlstm_lazy_cell_kernelILi3ELi32ELi16ELin1E")          # T = 3, 128-column slices, scores from the score kernel
    assert 171 <= sliced["VGPRs"] <= 256 and sliced["AGPRs"] == 0, sliced        # two workgroups per CU, by registers
    assert sliced["VGPRs Spill"] == 0
    wide = _one(res, "mlstm_lazy_cell_kernelILi3ELi64ELi8ELi4E")                 # T = 3, 256-wide heads (the headline kernel)
    assert wide["VGPRs"] <= 128 and wide["VGPRs Spill"] == 0, wide


def test_projection_workgroups_fit_beside_two_read_passes():
    res = _resources("gemm_f16x2p.hip")
    two_stage = _one(res, "gemm_f16x2p_kernelILb0ELb0ELi2ELi2ELi2ELi2E")         # 128 x 128 tile, two LDS stages
    assert two_stage["VGPRs"] <= 128 and two_stage["LDS Size [bytes/block]"] <= 68 * 1024, two_stage
    one_stage = _one(res, "gemm_f16x2p_kernelILb0ELb0ELi1ELi2ELi2ELi2E")         # 128 x 128 tile, one stage: four per CU
    assert one_stage["VGPRs"] <= 128 and one_stage["LDS Size [bytes/block]"] <= 36 * 1024, one_stage


def test_on_the_fly_projection_workgroup_budget():
    res = _resources("gemm_f16x2.hip")
    two_stage = _one(res, "gemm_f16x2_kernelILb0ELb1ELi64ELb0ELi2ELb1E")        # proj_down of chain-bound slices: 64-row tile, two stages
    assert two_stage["VGPRs"] <= 160 and two_stage["LDS Size [bytes/block]"] <= 48 * 1024, two_stage
    gated = _one(res, "gemm_f16x2_kernelILb0ELb1ELi64ELb1ELi1ELb0E")            # proj_down of the headline: 64-row tile, gate in the staging
    assert gated["VGPRs"] <= 128 and gated["LDS Size [bytes/block]"] <= 36 * 1024, gated
