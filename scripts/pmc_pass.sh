#!/bin/bash
# HBM traffic of the dominant kernel from rocprofv3 PMC counters, one counter per pass
# (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2 -> separate passes; no trace domains mixed in).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 $R/bench.py --steps 26 --warmup 4 \
      --no-cpu-baseline --no-kernel-timing --host-io-steps 0 ${BENCH_ARGS:-} > $OUT/$C.json 2> $OUT/$C.err
  echo "$C rc=$?"
done
cd $R && python3 scripts/parse_pmc.py $OUT ${PMC_TAG:-xlstm_16m} ${PMC_BATCH:-4096} ${PMC_MICRO:-2} ${PMC_FIRST:-20}
