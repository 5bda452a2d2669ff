#!/usr/bin/env python3
"""Whole-step HBM traffic by kernel family from the rocprofv3 --pmc passes of scripts/pmc_pass.sh (mean over the
steady-state env-steps of the two-slice pipeline: steps >= 20 of 16 priming + 4 warm-up + 26 timed, i.e. two full fold
periods): FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE, in GB per step.

    python scripts/parse_pmc_step.py gpurun_out/pmc <ms_per_step> [first_steady_step] > profiles/<round>_whole_step_hbm_traffic.json

The x2 read correction is exact for 16-byte-per-lane streaming loads (the state pass, norms, GEMM staging); kernels that
read with narrower lane accesses (fold tiles, window columns) are over-corrected by it, so the total is an upper bound."""
import collections, csv, glob, json, re, sys

out_dir, ms_per_step = sys.argv[1], float(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 20
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"{out_dir}/{counter}/**/*counter_collection.csv", recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    names = [r["Kernel_Name"] for r in rows]
    ends = [i for i, n in enumerate(names) if "action_argmax" in n][1::2]   # two slices -> two argmax per step
    ends = ends[:first + 26]                                                 # two fold periods; nothing past the timed region
    lo, hi = ends[first - 1] + 1, ends[-1] + 1                               # steady state: every step from `first` on
    n_steps = len(ends) - first
    fam = collections.Counter()
    for r in rows[lo:hi]:
        n = r["Kernel_Name"]
        n = n[5:] if n.startswith("void ") else n
        n = re.sub(r"\(.*$", "", n.replace("lram::(anonymous namespace)::", "").replace("lram::", ""))
        fam[re.sub(r"<.*$", "", n)] += float(r["Counter_Value"]) / n_steps
    res[counter] = (fam, (hi - lo) / n_steps)
fams = sorted(set(res["FETCH_SIZE"][0]) | set(res["WRITE_SIZE"][0]),
              key=lambda k: -(res["FETCH_SIZE"][0][k] * 2 + res["WRITE_SIZE"][0][k]))
table, tot = [], 0.0
for k in fams:
    rd, wr = res["FETCH_SIZE"][0][k] * 1024 * 2 / 1e9, res["WRITE_SIZE"][0][k] * 1024 / 1e9
    tot += rd + wr
    table.append({"kernel": k, "read_GB": round(rd, 3), "write_GB": round(wr, 3)})
print(json.dumps({"config": "xlstm_16m, 4096 env slots, lazy state, two-slice pipeline", "kernels_in_step": res["FETCH_SIZE"][1], "steps_averaged": n_steps,
                  "per_kernel_family": table, "total_GB_per_step": round(tot, 2), "ms_per_step": ms_per_step,
                  "average_GBps": round(tot / ms_per_step * 1e3, 0), "frac_of_8TBps": round(tot / ms_per_step / 8.0, 3),
                  "corrections": "FETCH_SIZE KiB x1024 x2, WRITE_SIZE KiB x1024 (upper bound: narrow reads over-corrected)"},
                 indent=1))
