"""Split rocprofv3's per-kernel average for mlstm_cell_kernel by launch shape and set it beside the live HIP-event
figures bench.py printed in the same profiled command.

    python scripts/summarize_cell_trace.py gpurun_out/prof_headline > profiles/<round>_cell_kernel_rocprof_vs_live.json

The kernel-stats CSV averages two launch shapes together (the pipelined half-batch launches of the timed region
and the full-batch launches of bench.py's standalone measurement); the kernel trace separates them by grid size.
"""
import collections
import csv
import glob
import json
import sys


def main(prof_dir: str) -> None:
    trace = glob.glob(f"{prof_dir}/**/*kernel_trace.csv", recursive=True)[0]
    shapes = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "mlstm_cell_kernel" in r["Kernel_Name"]:
            shapes[int(r["Grid_Size_Z"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    live = json.loads(open(f"{prof_dir}/bench.json").readline())
    roof = live["roofline"]
    per_env = roof["algorithmic_bytes_per_launch"] / (live["config"]["batch_per_gpu"] / 2)
    out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 4",
           "kernel": "mlstm_cell_kernel<3, 64, 16>", "algorithmic_bytes_per_env_per_launch": per_env, "shapes": []}
    for envs, d in sorted(shapes.items()):
        avg_ms = sum(d) / len(d) / 1e6
        is_pipe = envs != live["config"]["batch_per_gpu"]
        live_ms = roof["avg_launch_ms"] if is_pipe else roof["standalone"]["avg_launch_ms"]
        out["shapes"].append({"envs_per_launch": envs, "launches": len(d), "rocprof_avg_ms": round(avg_ms, 4),
                              "bench_live_avg_ms": round(live_ms, 4), "rocprof_over_live": round(avg_ms / live_ms, 4),
                              "rocprof_GBps": round(per_env * envs / avg_ms / 1e6, 1),
                              "role": "timed region (micro-batch pipeline)" if is_pipe else "standalone measurement"})
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
