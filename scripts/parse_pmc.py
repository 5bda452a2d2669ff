#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to per-launch HBM bytes of the recurrent kernel, STEADY STATE ONLY.

    python scripts/parse_pmc.py <pmc_dir> <config tag> <env slots> <env slices> [first_steady_step]

The run behind it (scripts/pmc_pass.sh) is `bench.py --steps 26 --warmup 4`: 16 priming + 4 warm-up + 26 timed steps.
Every env starts with an empty matrix memory and first folds within one fold period (13 steps), so until step 13 part
of the envs skip the C_base read and the priming folds have no C_base to read either.  Launches are therefore cut by
env-step (dispatches in Dispatch_Id order, a step ends with its last `action_argmax_kernel`) and only steps
>= `first_steady_step` (default 20 = priming + warm-up; two full fold periods, 26 steps, follow) are kept:
mean bytes per read launch + (all fold bytes of those steps) / (read launches of those steps).  Round 3's reducer took
the MEDIAN launch of a run that was 80 % priming steps and under-reported the read pass by 23 %.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports half the
bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane stores.
The library's stream_copy_kernel (1 GiB read + 1 GiB written, same access width) runs in the same process
and calibrates both factors."""
import csv
import glob
import json
import os
import sys

CELL = ("mlstm_cell_kernel", "mlstm_lazy_cell_kernel", "mamba_ssm_kernel", "mamba_ssm_lane_kernel")
FOLD = ("mlstm_lazy_fold_kernel",)
COPY = ("stream_copy",)
N_STEADY = int(os.environ.get("PMC_STEADY_STEPS", "26"))   # two fold periods


def steady_rows(rows, n_micro, first_step):
    """rows sorted by dispatch; returns (rows of steps >= first_step, number of such steps).  One env-step ends with the
    n_micro-th `action_argmax_kernel` since the previous step's end (one argmax launch per env slice)."""
    ends, seen = [], 0
    for i, r in enumerate(rows):
        if "action_argmax" in r["Kernel_Name"]:
            seen += 1
            if seen % n_micro == 0:
                ends.append(i)
    if len(ends) <= first_step:
        raise SystemExit(f"parse_pmc: only {len(ends)} env-steps in the trace, need more than {first_step}")
    ends = ends[:first_step + N_STEADY]          # anything bench.py runs after its timed region is not the steady state
    lo = ends[first_step - 1] + 1 if first_step > 0 else 0
    hi = ends[-1] + 1
    return rows[lo:hi], len(ends) - first_step


def reduce(out_dir, n_micro, first_step):
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(os.path.join(out_dir, counter, "**", "*counter_collection.csv"), recursive=True))
        if not files:
            raise SystemExit(f"parse_pmc: no counter_collection.csv under {out_dir}/{counter}")
        rows = [r for r in csv.DictReader(open(files[-1])) if r.get("Counter_Name") == counter]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        copy = [float(r["Counter_Value"]) for r in rows if any(n in r["Kernel_Name"] for n in COPY)]
        kept, n_steps = steady_rows(rows, n_micro, first_step)
        cell = [float(r["Counter_Value"]) for r in kept if any(n in r["Kernel_Name"] for n in CELL)]
        fold = [float(r["Counter_Value"]) for r in kept if any(n in r["Kernel_Name"] for n in FOLD)]
        lazy = any("mlstm_lazy_cell_kernel" in r["Kernel_Name"] for r in kept)
        res[counter] = {"cell_mean_KiB": sum(cell) / max(len(cell), 1), "cell_min_KiB": min(cell) if cell else None,
                        "cell_max_KiB": max(cell) if cell else None, "cell_launches": len(cell),
                        "fold_total_KiB": sum(fold), "fold_launches": len(fold),
                        "fold_share_KiB": sum(fold) / max(len(cell), 1),
                        "copy_KiB": sorted(copy)[len(copy) // 2] if copy else None, "steps": n_steps, "lazy": lazy}
    return res


def main(argv):
    out_dir, tag, batch = argv[1], argv[2], int(argv[3])
    n_micro = int(argv[4]) if len(argv) > 4 else 1   # env slices per step: one launch covers batch / n_micro envs
    first_step = int(argv[5]) if len(argv) > 5 else 20
    res = reduce(out_dir, n_micro, first_step)
    f, w = res["FETCH_SIZE"], res["WRITE_SIZE"]
    lazy = bool(f["lazy"])
    GiB = 1024 ** 3
    rd = f["cell_mean_KiB"] * 1024 * 2.0
    wr = w["cell_mean_KiB"] * 1024
    summary = {"config": tag, "batch": batch, "micro_batches": n_micro, "envs_per_launch": batch // n_micro,
               "state_mode": "lazy" if lazy else "materialised",
               "reduction": f"mean over the launches of env-steps >= {first_step} of the run ({f['steps']} steady-state "
                            f"steps, {f['cell_launches']} read launches, {f['fold_launches']} fold launches); priming and "
                            "warm-up steps dropped",
               "raw": res, "read_pass_alone": {"read_bytes": rd, "write_bytes": wr}}
    if lazy:
        # the fold reads C_base tiles and window rows in 16-byte lane runs as well: the same x2 applies (round 3 took its
        # FETCH_SIZE at face value, which was a priming-run artefact: folds without a C_base to read)
        frd = f["fold_share_KiB"] * 1024 * 2.0
        fwr = w["fold_share_KiB"] * 1024
        summary["fold_share_per_read_launch"] = {"read_bytes": frd, "write_bytes": fwr}
        rd += frd
        wr += fwr
    summary.update(read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
                   hbm_bytes_per_env_per_launch=(rd + wr) / (batch // n_micro),
                   corrections="FETCH_SIZE KiB x1024 x2 (gfx950 wide-read undercount), WRITE_SIZE KiB x1024")
    if f["copy_KiB"] and w["copy_KiB"]:
        summary["calibration_stream_copy"] = {"fetch_reported_over_true": f["copy_KiB"] * 1024 / GiB,
                                              "write_reported_over_true": w["copy_KiB"] * 1024 / GiB}
    os.makedirs("profiles", exist_ok=True)
    rnd = os.environ.get("PMC_ROUND", "r04")
    path = f"profiles/{rnd}_cell_kernel_hbm_traffic{'' if tag == 'xlstm_16m' else '_' + tag}{'_lazy' if lazy else ''}.json"
    json.dump(summary, open(path, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(sys.argv)
