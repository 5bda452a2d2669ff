"""GPU: bench.py's own line -- the contract keys, the physical roofline (fractions <= 1), the host-inclusive leg -- on a
reduced batch so that it runs in seconds."""
import pytest

pytestmark = pytest.mark.gpu


def test_bench_line_contract_and_physical_roofline(hip_lib, capsys):
    import bench
    out = bench.main(["--config", "xlstm_16m", "--batch", "512", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                      "--no-stream-ceilings", "--host-io-steps", "3", "--timing-every", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["metric"].startswith("env-steps/sec") and out["unit"] == "env-steps/s" and out["dtype"] == "f32"
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["warmup"] == 2 and out["vs_baseline"] is None
    assert out["config"]["state_mode"] == "lazy"
    # the workload line names the projection kernel that actually served most of the FLOPs (engine dispatch counters), not
    # the LRAM_GEMM default: at 512 env slots the two slices hold 768 operand rows each, above the f16x2 kernels' 256-row
    # threshold (1024 until round 6: bf16x3 then) -- and LRAM_F16_MIN_ROWS=2048 puts the same run on bf16x3
    disp = out["gemm_dispatch_per_step"]
    main = max(disp, key=lambda k: disp[k]["gflop"])
    assert main == "f16x2" and "f16x2" in out["config"]["workload"] and disp[main]["launches"] > 0
    assert out["ranks_seen"] == 1
    assert abs(out["value"] - 512 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.0 < r["standalone"]["frac"] <= 1.0
    # 7 mLSTM blocks x 2 slices; 7 folds; every 2nd of the 4 timed steps carries the event pairs (lram_profile_begin_sampled)
    assert r["launches_per_step"] == 14 and r["steps_timed"] == 2 and r["fold_launches_timed"] == 7 * 2 and r["launches_timed"] == 28
    assert r["effective_8d_GBps"] > r["achieved"]                                   # the 8d figure prices more bytes
    assert (r["traffic"] is None) == (r["traffic_source"] is None)                  # a replayed constant is labelled
    if r["traffic"] is not None:                                                    # ... and agrees with the byte model
        assert 0.95 <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= 1.10
    h = out["host_io"]
    assert h["steps"] == 3 and h["value"] > 0 and h["bytes_h2d_per_step"] == 512 * (204 * 4 + 4 + 1)
    # both definitions of the metric on the line, under names that say which is which: `value` = inputs resident in HBM
    # (the bench contract), `value_host_inclusive` = SURVEY 8d's H2D + step + D2H + host sync
    assert out["value_inputs_in_hbm"] == out["value"] and out["value_host_inclusive"] == h["value"]
    # the library that ran is the one the checked-out sources build (sha256 over sources + headers + flags, compiled in)
    from lram_amd import build
    assert out["build_id"] == build.source_hash() and out["build_id_matches_sources"] is True
    assert "cpu_baseline" not in out and "_copy_ceiling_pending" not in out
    capsys.readouterr()


def test_bench_config_legs_contract(hip_lib, capsys):
    """`configs` on the default bench line: BASELINE.json's configurations 2-5 at their per-GPU sizes, each timed on a fresh
    engine with its own roofline / matrix-core block (bench.config_legs).  Here: the contract of every leg, fractions physical,
    the reference Mamba trajectory slower than the one-advance-per-step one by about its 4 forwards."""
    import bench
    legs = bench.config_legs("cuda:0")
    assert [l.get("id") for l in legs] == ["C2", "C3", "C3-reference-trajectory", "C4-per-gpu-shard", "C5"], legs
    for l in legs:
        assert "error" not in l, l
        for key in ("workload", "preset", "batch", "steps", "value", "unit", "ms_per_step", "mfma", "leg_wall_s"):
            assert key in l, (l["id"], key)
        assert l["value"] > 0 and l["ms_per_step"] > 0 and l["leg_wall_s"] < 60
        assert 0.0 < l["mfma"]["frac"] <= 1.0 and l["mfma"]["peak"] == 2.5 and l["mfma"]["unit"] == "PFLOP/s"
        if l["id"] != "C5":
            assert l["unit"] == "env-steps/s"
            assert abs(l["value"] - l["batch"] * 1e3 / l["ms_per_step"]) < 1e-6 * l["value"]
            r = l["roofline"]
            assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] <= 1.0
            assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["launches_timed"] > 0
    by = {l["id"]: l for l in legs}
    assert by["C2"]["state_mode"] == "lazy" and by["C4-per-gpu-shard"]["state_mode"] == "lazy"
    assert by["C2"]["roofline"]["launches_per_step"] == 14 and by["C4-per-gpu-shard"]["roofline"]["launches_per_step"] == 34
    ratio = by["C3"]["value"] / by["C3-reference-trajectory"]["value"]
    assert 3.0 < ratio < 5.0, ratio                  # 4 forwards per env-step on the reference's Meta-World trajectory
    c5 = by["C5"]
    assert c5["unit"] == "env-timesteps/s" and c5["context_timesteps"] == 512 and c5["batch"] == 64
    assert abs(c5["value"] - 64 * 512 * 1e3 / c5["ms_per_step"]) < 1e-6 * c5["value"]
    assert c5["decode"]["steps"] == 16 and c5["decode"]["value"] > 0
    # the chunkwise state pass's own roofline: 17 mLSTM blocks x 25 chunks of the 512-timestep context, timed by live events
    r5 = c5["roofline"]
    assert r5["bound"] == "hbm" and r5["peak"] == 8000.0 and 0.0 < r5["frac"] <= 1.0 and r5["launches_timed"] == 17 * 25
    assert abs(r5["frac"] - r5["achieved"] / r5["peak"]) < 1e-12 and abs(r5["state_passes_per_block"] - 25) < 1e-9
    assert r5["standalone"]["launches_timed"] == 17 * 25 and r5["frac"] <= r5["standalone"]["frac"] <= 1.0   # (one chunk at a time)
    capsys.readouterr()
