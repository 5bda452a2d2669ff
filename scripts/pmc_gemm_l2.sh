#!/bin/bash
# L2 hit rate / beyond-L2 traffic and stall split of the projection kernels per shape (standalone launches of scripts/bench_gemm.py):
# two rocprofv3 --pmc passes (TCC block: 4 slots; SQ block: 8), counters only beside --kernel-trace.
#   usage: pmc_gemm_l2.sh <kernels...>      -> gpurun_out/pmc_gemm_l2/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_gemm_l2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -- python3 $R/scripts/bench_gemm.py "$@" > $OUT/order.json 2> $OUT/err1.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/scripts/bench_gemm.py "$@" > /dev/null 2> $OUT/err2.log
python3 - $OUT <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
order = json.loads(open(out + "/order.json").readline())
def load(sub):
    rows = collections.defaultdict(dict)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm_" in n and "splitk" not in n:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    return rows, sorted(rows)
tcc, tids = load("tcc")
sq, sids = load("sq")
i = 0
lines = []
for o in order:
    t = tcc[tids[i + o["reps"] - 1]]
    s = sq[sids[i + o["reps"] - 1]]
    i += o["reps"]
    hit = t["TCC_HIT_sum"] / max(t["TCC_HIT_sum"] + t["TCC_MISS_sum"], 1)
    beyond = t["TCC_EA0_RDREQ_sum"] * 64 * 2 / 1e6      # MB (64 B per request as tallied, x2: gfx950 wide-read undercount)
    unique = (o["m"] + o["n"]) * o["k"] * 4 / 1e6
    wc = s["SQ_WAVE_CYCLES"]
    busy = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * s["GRBM_GUI_ACTIVE"])
    lines.append(f"{o['tag']:14s} {o['kernel']:7s} L2 hit {hit:5.3f}  beyond-L2 reads {beyond:8.1f} MB (operands {unique:6.1f} MB)  "
                 f"parked {s['SQ_WAIT_ANY'] / wc:5.3f} issue-stall {s['SQ_WAIT_INST_ANY'] / wc:5.3f} issuing {s['SQ_ACTIVE_INST_ANY'] / wc:5.3f}  "
                 f"mfma_busy {busy:5.3f}  valu/wave {s['SQ_INSTS_VALU'] / max(s['SQ_WAVES'], 1):7.0f}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
