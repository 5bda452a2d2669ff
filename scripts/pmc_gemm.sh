#!/bin/bash
# SQ counters of the GEMM kernels over scripts/bench_gemm.py's shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/bench_gemm.py "$@" > $OUT/order.json 2> $OUT/err.log
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
order = json.loads(open(sys.argv[1] + "/order.json").readline())
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16x3_kernel" in r["Kernel_Name"] or "gemm_f32_kernel" in r["Kernel_Name"] or "gemm_f16x2_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
i = 0
for o in order:
    d = rows[ids[i + o["reps"] - 1]]
    i += o["reps"]
    wc = d["SQ_WAVE_CYCLES"]
    print(f"{o['tag']:14s} {o['kernel']:7s} wait_any {d['SQ_WAIT_ANY']/wc:5.2f} wait_inst {d['SQ_WAIT_INST_ANY']/wc:5.2f} (lds {d['SQ_WAIT_INST_LDS']/wc:5.2f}) active {d['SQ_ACTIVE_INST_ANY']/wc:5.2f}  lds_conflict/active {d['SQ_LDS_BANK_CONFLICT']/max(d['SQ_LDS_IDX_ACTIVE'],1):5.2f}  mfma_busy/wave_cyc {d['SQ_VALU_MFMA_BUSY_CYCLES']/(4*wc):5.2f}")
PY
