#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to per-launch HBM bytes of the recurrent kernel.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports half the
bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane stores.
The library's stream_copy_kernel (1 GiB read + 1 GiB written, same access width) runs in the same process
and calibrates both factors."""
import csv
import glob
import json
import os
import sys

out_dir, tag, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
n_micro = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # env slices per step: one launch covers batch / n_micro envs
KERNELS = {"cell": ("mlstm_cell_kernel", "mlstm_lazy_cell_kernel", "mamba_ssm_kernel"), "copy": ("stream_copy",),
           "fold": ("mlstm_lazy_fold_kernel",)}
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out_dir, counter, "**", "*counter_collection.csv"), recursive=True)
    vals = {"cell": [], "copy": [], "fold": []}
    lazy = False
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            for key, names in KERNELS.items():
                if any(n in row["Kernel_Name"] for n in names):
                    vals[key].append(float(row["Counter_Value"]))
                    lazy = lazy or "mlstm_lazy_cell_kernel" in row["Kernel_Name"]
    # median = steady-state launch (the first timestep resets every env and skips the C read)
    res[counter] = {k: (sorted(v)[len(v) // 2] if v else None, len(v)) for k, v in vals.items()}
    # lazy matrix memory: a state pass = one read launch + its share of the fold launches (all fold bytes of the run
    # spread over the read launches; a fold launch covers every slice and ~1/13 of the envs)
    res[counter]["fold_mean"] = (sum(vals["fold"]) / max(len(vals["cell"]), 1) if vals["fold"] else 0.0, len(vals["fold"]))
    res[counter]["lazy"] = lazy
print(res)
GiB = 1024 ** 3
fetch_cell, nfc = res["FETCH_SIZE"]["cell"]
write_cell, nwc = res["WRITE_SIZE"]["cell"]
fetch_copy, _ = res["FETCH_SIZE"]["copy"]
write_copy, _ = res["WRITE_SIZE"]["copy"]
summary = {"config": tag, "batch": batch, "micro_batches": n_micro, "envs_per_launch": batch // n_micro, "raw_KiB_median_per_launch": {k: {kk: vv[0] for kk, vv in v.items() if isinstance(vv, tuple)} for k, v in res.items()},
           "launches": {"fetch": nfc, "write": nwc}}
lazy = bool(res["FETCH_SIZE"].get("lazy"))
summary["state_mode"] = "lazy" if lazy else "materialised"
if fetch_cell is not None and write_cell is not None:
    rd = fetch_cell * 1024 * 2.0
    wr = write_cell * 1024
    if lazy:
        # the fold kernel reads C with 4-byte lane accesses (no wide-read undercount on those) and k/v rows with 16-byte
        # ones: its FETCH_SIZE is taken at face value here, i.e. as a lower bound
        summary["fold_share_per_read_launch"] = {"read_bytes_min": res["FETCH_SIZE"]["fold_mean"][0] * 1024,
                                                 "write_bytes": res["WRITE_SIZE"]["fold_mean"][0] * 1024}
        rd += res["FETCH_SIZE"]["fold_mean"][0] * 1024
        wr += res["WRITE_SIZE"]["fold_mean"][0] * 1024
    summary.update(read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
                   hbm_bytes_per_env_per_launch=(rd + wr) / (batch // n_micro),
                   corrections="FETCH_SIZE KiB x1024 x2 (gfx950 wide-read undercount), WRITE_SIZE KiB x1024")
    if fetch_copy and write_copy:
        summary["calibration_stream_copy"] = {"fetch_reported_over_true": fetch_copy * 1024 / GiB,
                                              "write_reported_over_true": write_copy * 1024 / GiB}
os.makedirs("profiles", exist_ok=True)
rnd = os.environ.get("PMC_ROUND", "r02")
path = f"profiles/{rnd}_cell_kernel_hbm_traffic{'' if tag == 'xlstm_16m' else '_' + tag}{'_lazy' if lazy else ''}.json"
json.dump(summary, open(path, "w"), indent=1)
print(json.dumps(summary, indent=1))
