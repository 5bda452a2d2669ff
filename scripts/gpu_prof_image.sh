#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_image; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/bench_image_encoder.py "$@" > $OUT/out.log 2> $OUT/err.log
cat $OUT/out.log
python3 - $OUT <<'PY'
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "conv3x3" in n or "maxpool" in n or "relu_kernel" in n or "gemm" in n:
        acc[(n[:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["LDS_Block_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), round(sum(v) / len(v) / 1e3, 1), "us")
PY
