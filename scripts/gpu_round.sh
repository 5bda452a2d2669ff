#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel-trace summary of the same bench command, PMC pass,
# and the secondary measurements quoted in DESIGN.md (Mamba, prefill, image front end, GEMM micro-benchmark).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log | cut -c1-200
python bench.py --steps ${STEPS:-64} --warmup 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cut -c1-400 $OUT/bench.json; tail -2 $OUT/bench.err | cut -c1-200
python bench.py --steps ${STEPS:-64} --warmup 8 --state eager --no-cpu-baseline > $OUT/bench_eager.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_eager.json
bash scripts/gpu_prof.sh headline --steps 16 --warmup 4
bash scripts/gpu_prof.sh headline_eager --steps 16 --warmup 4 --state eager | head -6
rm -rf $OUT/pmc; bash scripts/pmc_pass.sh 2>&1 | grep -E "rc=|hbm_bytes_per_env|state_mode"; rm -rf $OUT/pmc_lazy; mv $OUT/pmc $OUT/pmc_lazy
BENCH_ARGS="--state eager" bash scripts/pmc_pass.sh 2>&1 | grep -E "rc=|hbm_bytes_per_env|state_mode"; rm -rf $OUT/pmc_eager; mv $OUT/pmc $OUT/pmc_eager
if [ "${FULL:-1}" = "1" ]; then
  bash scripts/gpu_prof.sh mamba --config mamba_48m --batch 2048 --steps 16 --warmup 4 | head -8
  PREFILL_MODES=chunkwise bash scripts/gpu_prof_prefill.sh xlstm_16m 512 63 1 | head -10
  python scripts/bench_prefill.py xlstm_16m 1024 63 > $OUT/prefill_16m.txt 2>/dev/null; cat $OUT/prefill_16m.txt
  python scripts/bench_prefill.py xlstm_206m 64 512 > $OUT/prefill_206m.txt 2>/dev/null; cat $OUT/prefill_206m.txt
  python scripts/bench_image_encoder.py 512 1280 > $OUT/image_encoder.txt 2>/dev/null; cat $OUT/image_encoder.txt
  bash scripts/gpu_gemm.sh bf16x3 f32 > $OUT/gemm_micro.txt 2>/dev/null; head -4 $OUT/gemm_micro.txt
  bash scripts/gpu_sweep.sh > $OUT/sweep.txt 2>/dev/null; cat $OUT/sweep.txt
fi
