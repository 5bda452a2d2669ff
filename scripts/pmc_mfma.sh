#!/bin/bash
# Matrix-pipe busy share of the projection kernels per shape: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs:
# 32 per v_mfma_f32_32x32x16_*, MI355X_MICROARCH.md) over 1024 SIMDs x the kernel's cycles (GRBM_GUI_ACTIVE is summed
# over the 8 XCDs).  Counters only, no trace domains beside --kernel-trace.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_mfma; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/bench_gemm.py f16x2 f16x2p bf16x3 f32 > $OUT/order.json 2> $OUT/err.log
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
order = json.loads(open(sys.argv[1] + "/order.json").readline())
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if ("gemm_bf16x3_kernel" in n or "gemm_f32_kernel" in n or "gemm_f16x2_kernel" in n or "gemm_f16x2p_kernel" in n or "gemm_f16x2_8p_kernel" in n):
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
i = 0
out = collections.defaultdict(dict)
for o in order:
    d = rows[ids[i + o["reps"] - 1]]
    i += o["reps"]
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * d["GRBM_GUI_ACTIVE"])
    out[o["kernel"]][o["tag"]] = round(busy, 4)
    print(f"{o['tag']:14s} {o['kernel']:7s} M={o['m']:6d} N={o['n']:5d} K={o['k']:5d}  mfma_busy {busy:6.3f}")
json.dump({"what": "matrix-pipe busy share of the projection kernel per shape: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel "
                   "cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (scripts/pmc_mfma.sh, standalone launches of scripts/bench_gemm.py)",
           **out}, open(sys.argv[1] + "/mfma_busy.json", "w"), indent=1)
PY
