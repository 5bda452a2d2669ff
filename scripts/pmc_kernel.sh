#!/bin/bash
# SQ stall / issue counters of ONE kernel family inside a bench.py run (standalone launches: --micro 1), one rocprofv3 --pmc
# pass, program directly after `--` (MI355X_MICROARCH.md: 8 SQ slots + 2 GRBM per pass; no trace domains beside --kernel-trace).
#   usage: pmc_kernel.sh <tag> <kernel name substring> <bench.py args ...>     -> gpurun_out/pmc_<tag>/summary.json
# WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; KERN=$2; shift 2
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --host-io-steps 0 \
  --no-stream-ceilings --micro 1 --steps 6 --warmup 2 "$@" > $OUT/bench.json 2> $OUT/err.log
python3 - $OUT "$KERN" <<'PY'
import collections, csv, glob, json, sys
out, kern = sys.argv[1], sys.argv[2]
per = collections.defaultdict(dict)
dur = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ids = sorted(per)[len(per) // 2:]          # steady state: the second half of the launches
mean = lambda k: sum(per[i].get(k, 0.0) for i in ids) / max(len(ids), 1)
wc = mean("SQ_WAVE_CYCLES")
s = {"kernel": kern, "launches_averaged": len(ids), "avg_us_under_pmc": sum(dur[i] for i in ids if i in dur) / max(len(ids), 1) / 1e3,
     "waves": mean("SQ_WAVES"), "wave_quad_cycles": wc,
     "share_parked_on_waitcnt_or_barrier": mean("SQ_WAIT_ANY") / wc, "share_issue_stalled": mean("SQ_WAIT_INST_ANY") / wc,
     "share_issuing": mean("SQ_ACTIVE_INST_ANY") / wc, "share_issuing_valu": mean("SQ_ACTIVE_INST_VALU") / wc,
     "valu_insts_per_wave": mean("SQ_INSTS_VALU") / max(mean("SQ_WAVES"), 1), "smem_insts_per_wave": mean("SQ_INSTS_SMEM") / max(mean("SQ_WAVES"), 1),
     # mean resident waves per SIMD while the kernel runs: wave cycles (x4: quad-cycles) / (1024 SIMDs x kernel cycles)
     "mean_waves_per_simd": 4.0 * wc / (1024.0 * mean("GRBM_GUI_ACTIVE") / 8.0) if mean("GRBM_GUI_ACTIVE") else None}
json.dump(s, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY
