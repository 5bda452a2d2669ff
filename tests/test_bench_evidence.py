"""CPU: the committed counter evidence agrees with bench.py's byte model of the dominant kernel.

`roofline.traffic` on the bench line is a constant replayed from profiles/rNN_cell_kernel_hbm_traffic_lazy.json (a separate
rocprofv3 --pmc pass, steady-state launches only: scripts/parse_pmc.py).  If the byte model of the state pass and the HBM
counters drift apart -- a kernel starts re-reading, or the reducer picks the wrong launches as round 3's did -- this fails."""
import glob
import json
import os

import bench
from lram_amd import preset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_state_pass_counters_match_the_byte_model():
    pm = json.load(open(_latest("r0[4-9]_cell_kernel_hbm_traffic_lazy.json")))
    spec = preset(pm["config"])
    model = bench.cell_bytes_lazy(spec, spec.tokens_per_step)
    ratio = pm["hbm_bytes_per_env_per_launch"] / model
    assert 0.95 <= ratio <= 1.10, (ratio, pm["hbm_bytes_per_env_per_launch"], model)
    assert "steady-state" in pm["reduction"]
    cal = pm["calibration_stream_copy"]   # the x2 read correction and the exact write counter, calibrated in the same run
    assert abs(cal["fetch_reported_over_true"] - 0.5) < 0.01 and abs(cal["write_reported_over_true"] - 1.0) < 0.01


def test_state_pass_file_agrees_with_the_whole_step_reduction():
    """Two reductions of one PMC run (per-launch mean of the state pass, per-family sums of the whole step) within 3 %."""
    pm = json.load(open(_latest("r0[4-9]_cell_kernel_hbm_traffic_lazy.json")))
    ws = json.load(open(_latest("r0[4-9]_whole_step_hbm_traffic.json")))
    fam = {r["kernel"]: r["read_GB"] + r["write_GB"] for r in ws["per_kernel_family"]}
    spec = preset(pm["config"])
    launches = (spec.n_blocks - len(spec.slstm_at)) * pm["micro_batches"]
    per_launch = (fam["mlstm_lazy_cell_kernel"] + fam["mlstm_lazy_fold_kernel"]) * 1e9 / launches
    assert abs(pm["hbm_bytes_per_launch"] / per_launch - 1.0) < 0.03


def test_mamba_state_update_counters_match_the_byte_model():
    """Same check for the C3 line's dominant kernel (the selective state update): its `roofline.traffic` constant is replayed
    from profiles/rNN_cell_kernel_hbm_traffic_mamba_48m.json (scripts/pmc_pass.sh with PMC_TAG=mamba_48m)."""
    pm = json.load(open(_latest("r0[4-9]_cell_kernel_hbm_traffic_mamba_48m.json")))
    spec = preset(pm["config"])
    model = bench.ssm_bytes(spec, spec.tokens_per_step)
    ratio = pm["hbm_bytes_per_env_per_launch"] / model
    assert 0.95 <= ratio <= 1.10, (ratio, pm["hbm_bytes_per_env_per_launch"], model)
    assert "steady-state" in pm["reduction"] and pm["envs_per_launch"] * pm["micro_batches"] == pm["batch"]
