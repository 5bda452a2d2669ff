"""RCCL through this repo's N > 1 code path on a ONE-GPU box.

No 8-GPU node has run this build yet (SCALE_r0N.json: skipped), so `lram_amd/dist.py` and bench.py's N > 1 branch have only
ever executed over gloo on CPU (tests/test_dist_rollout.py).  LRAM_DIST_SINGLE_RANK=1 makes a WORLD_SIZE == 1 job create its NCCL
(= RCCL on ROCm) process group bound to the device and issue every collective of the path -- barrier, all_gather_into_tensor of
the action tensor after every step, the all-reduces behind `ranks_seen` and the max-over-ranks timing, `collective_report` --
each the identity with one rank, but through RCCL's communicator and kernels.  Run in a child process: a process group is
process-global state."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_n_gt_1_path_runs_on_rccl_with_one_rank(hip_lib):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LRAM_DIST_SINGLE_RANK="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "512", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--no-stream-ceilings", "--host-io-steps", "3"]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out["value"] > 0
    c = out["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["library_version"]      # RCCL reports its version
    assert c["all_gather_bytes_per_rank"] == 512 * 8 * 4 and c["all_gather_us"] > 0.0
    assert out["host_io"]["value"] > 0                                                    # the host-inclusive leg gathers too
    assert "[bench] collective:" in proc.stderr
