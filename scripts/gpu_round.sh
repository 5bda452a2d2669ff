#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel-trace summary of the same bench command, PMC pass,
# and the secondary measurements quoted in DESIGN.md.  ROUND=r02 names the artifacts copied into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
RND=${ROUND:-r02}
mkdir -p $OUT $R/profiles
cd $R
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -14 $OUT/pytest_gpu.log | cut -c1-200
fi
python bench.py --steps ${STEPS:-64} --warmup 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cp $OUT/bench.json profiles/${RND}_bench_xlstm16m_b4096.json
python bench.py --steps ${STEPS:-64} --warmup 8 --state eager --no-cpu-baseline > $OUT/bench_eager.json 2>> $OUT/bench.err
cp $OUT/bench_eager.json profiles/${RND}_bench_xlstm16m_b4096_eager.json
cut -c1-330 $OUT/bench.json; echo; cut -c1-200 $OUT/bench_eager.json; echo
bash scripts/gpu_prof.sh headline --steps 16 --warmup 4 | head -12
f=$(find $OUT/prof_headline -name "*kernel_stats.csv" | head -1); cp $f profiles/${RND}_kernel_stats_xlstm16m_b4096.csv
python scripts/summarize_state_pass.py $OUT/prof_headline > profiles/${RND}_state_pass_rocprof_vs_live.json; cat profiles/${RND}_state_pass_rocprof_vs_live.json | head -40
t=$(find $OUT/prof_headline -name "*kernel_trace.csv" | head -1); python scripts/timeline.py $t -3 30 > profiles/${RND}_step_timeline_xlstm16m_b4096.txt
rm -rf $OUT/pmc; PMC_ROUND=$RND bash scripts/pmc_pass.sh 2>&1 | grep -E "rc=|hbm_bytes_per_env|state_mode"
python scripts/parse_pmc_step.py $OUT/pmc $(python -c "import json; print(json.load(open('$OUT/bench.json'))['ms_per_step'])") > profiles/${RND}_whole_step_hbm_traffic.json 2>/dev/null
if [ "${FULL:-0}" = "1" ]; then
  bash scripts/gpu_prof.sh mamba --config mamba_48m --batch 2048 --steps 16 --warmup 4 | head -8
  f=$(find $OUT/prof_mamba -name "*kernel_stats.csv" | head -1); cp $f profiles/${RND}_kernel_stats_mamba48m_b2048.csv
  bash scripts/gpu_sweep.sh > $OUT/sweep.txt 2>/dev/null; cp $OUT/sweep.txt profiles/${RND}_config_sweep.txt; cat $OUT/sweep.txt
fi
mkdir -p $OUT/profiles_out; cp profiles/${RND}_* $OUT/profiles_out/
