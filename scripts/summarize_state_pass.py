"""rocprofv3 per-kernel durations of the state pass (and the lazy mode's fold launches), split by launch shape, beside
the live HIP-event figures bench.py printed in the same profiled command.

    python scripts/summarize_state_pass.py gpurun_out/prof_<tag> > profiles/<round>_state_pass_rocprof_vs_live.json

The kernel-stats CSV averages two launch shapes together (the pipelined half-batch launches of the timed region and the
full-batch launches of bench.py's standalone measurement); the kernel trace separates them by grid size."""
import collections
import csv
import glob
import json
import sys


def main(prof_dir: str) -> None:
    trace = glob.glob(f"{prof_dir}/**/*kernel_trace.csv", recursive=True)[0]
    live = json.loads(open(f"{prof_dir}/bench.json").readline())
    roof = live["roofline"]
    B = live["config"]["batch_per_gpu"]
    cells, folds = collections.defaultdict(list), []
    name = None
    for r in csv.DictReader(open(trace)):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = r["Kernel_Name"]
        if "mlstm_lazy_cell_kernel" in k or "mlstm_cell_kernel" in k or "mamba_ssm_kernel" in k or "mamba_ssm_lane_kernel" in k:
            cells[int(r["Grid_Size_Z"])].append(d)
            name = k.split("(")[0].replace("void ", "").replace("lram::(anonymous namespace)::", "")
        elif "mlstm_lazy_fold_kernel" in k:
            folds.append(d)
    out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline " + live.get("argv", ""),
           "kernel": name, "bench_value_under_profiler": live["value"], "shapes": []}
    for envs, d in sorted(cells.items()):
        # the first launches of the run see empty matrix memories (nothing to read): steady state = the upper 3/4
        steady = sorted(d)[len(d) // 4:]
        avg_ms = sum(steady) / len(steady) / 1e6
        pipelined = envs != B
        r = roof if pipelined else roof.get("standalone", {})
        live_ms = r.get("state_pass_avg_ms", r.get("avg_launch_ms"))
        row = {"envs_per_launch": envs, "launches": len(d), "rocprof_avg_ms_steady": round(avg_ms, 4),
               "rocprof_avg_ms_all": round(sum(d) / len(d) / 1e6, 4),
               "role": "timed region (micro-batch pipeline)" if pipelined else "standalone measurement"}
        if live_ms:
            row["bench_live_avg_ms"] = round(live_ms, 4)
            row["rocprof_over_live"] = round(avg_ms / live_ms, 4)
        out["shapes"].append(row)
    if folds:
        steady = sorted(folds)[len(folds) // 4:]
        out["fold"] = {"launches": len(folds), "rocprof_avg_ms_steady": round(sum(steady) / len(steady) / 1e6, 4),
                       "bench_live_avg_ms_pipelined": round(roof.get("fold_avg_ms", 0.0), 4),
                       "bench_live_avg_ms_standalone": round(roof.get("standalone", {}).get("fold_avg_ms", 0.0), 4),
                       "note": "fold launches of the timed region and of the standalone measurement have the same grid"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
