#!/bin/bash
# SQ / TCC counters of the chunkwise prefill kernels (one rocprofv3 --pmc pass per counter group)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_chunk; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/scripts/bench_prefill.py ${1:-xlstm_16m} ${2:-512} ${3:-63} ${4:-0} > $OUT/p$i.log 2> $OUT/p$i.err
  echo "pass $i rc=$?"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("mlstm_cell_chunk_kernel", "mlstm_pre_tok_kernel", "mlstm_chunk_scan_kernel"):
            if name in k:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print(f"   {c:28s} median {v[len(v)//2]:.4g}  n={len(v)}")
PY
