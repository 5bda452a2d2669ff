#!/bin/bash
# per-kernel timeline of one env-step at a small batch (single slice): usage <tag> <bench args>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd $R; TAG=$1; shift
bash scripts/gpu_prof.sh sm "$@" --no-kernel-timing --host-io-steps 0 --no-stream-ceilings > /dev/null 2>&1
t=$(find $OUT/prof_sm -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $OUT/small_trace_$TAG.txt <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]; n = n[5:] if n.startswith("void ") else n
    r["name"] = re.sub(r"\(.*$", "", n.replace("lram::(anonymous namespace)::", "").replace("lram::", ""))[:60]
rows.sort(key=lambda r: r["s"])
arg = [i for i, r in enumerate(rows) if "action_argmax" in r["name"]]
lo, hi = arg[-3] + 1, arg[-2] + 1
step = rows[lo:hi]; t0 = step[0]["s"]
print(f"kernels {len(step)} span us {(step[-1]['e'] - t0) / 1e3:.1f} sum dur us {sum(r['e'] - r['s'] for r in step) / 1e3:.1f}")
prev = t0
for r in step:
    print(f"{(r['s'] - t0) / 1e3:8.1f} dur {(r['e'] - r['s']) / 1e3:6.1f} gap {(r['s'] - prev) / 1e3:5.1f} {r['name']}")
    prev = r["e"]
PY
rm -rf $OUT/prof_sm; head -1 $OUT/small_trace_$TAG.txt
