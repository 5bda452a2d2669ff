#!/bin/bash
# Round-6 artifact pass on one GPU box: full GPU suite, headline bench (+ eager), rocprofv3 kernel stats / state-pass
# summary / timeline of the same command, PMC traffic passes, Mamba-48M stats + bench lines, config sweep, MFMA busy table.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; RND=r06; mkdir -p $OUT/profiles_out; cd $R
if [ "${TESTS:-1}" = "1" ]; then
  LRAM_TEST_REPORT=1 timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -14 $OUT/pytest_gpu.log | cut -c1-200
fi
python bench.py --steps 64 --warmup 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cp $OUT/bench.json profiles/${RND}_bench_xlstm16m_b4096.json
python bench.py --steps 64 --warmup 8 --state eager --no-cpu-baseline > $OUT/bench_eager.json 2>> $OUT/bench.err
cp $OUT/bench_eager.json profiles/${RND}_bench_xlstm16m_b4096_eager.json
cut -c1-330 $OUT/bench.json; echo; cut -c1-200 $OUT/bench_eager.json; echo
bash scripts/gpu_prof.sh headline --steps 16 --warmup 4 | head -12
f=$(find $OUT/prof_headline -name "*kernel_stats.csv" | head -1); cp $f profiles/${RND}_kernel_stats_xlstm16m_b4096.csv
python scripts/summarize_state_pass.py $OUT/prof_headline > profiles/${RND}_state_pass_rocprof_vs_live.json; head -30 profiles/${RND}_state_pass_rocprof_vs_live.json
t=$(find $OUT/prof_headline -name "*kernel_trace.csv" | head -1); python scripts/timeline.py $t -3 30 > profiles/${RND}_step_timeline_xlstm16m_b4096.txt
python scripts/timeline.py $t -3 30 -v > profiles/${RND}_step_timeline_xlstm16m_b4096_verbose.txt
rm -rf $OUT/pmc; PMC_ROUND=$RND bash scripts/pmc_pass.sh 2>&1 | grep -E "rc=|hbm_bytes_per_env|state_mode"
python scripts/parse_pmc_step.py $OUT/pmc $(python -c "import json; print(json.load(open('$OUT/bench.json'))['ms_per_step'])") > profiles/${RND}_whole_step_hbm_traffic.json 2>/dev/null
bash scripts/gpu_prof.sh mamba --config mamba_48m --batch 2048 --steps 16 --warmup 4 | head -8
f=$(find $OUT/prof_mamba -name "*kernel_stats.csv" | head -1); cp $f profiles/${RND}_kernel_stats_mamba48m_b2048.csv
bash scripts/pmc_mfma.sh > $OUT/pmc_mfma.txt 2>&1; cat $OUT/pmc_mfma.txt | head -45; cp $OUT/pmc_mfma/mfma_busy.json profiles/${RND}_gemm_mfma_busy.json
bash scripts/pmc_gemm_l2.sh f16x2 f16x2p > $OUT/pmc_gemm_l2.txt 2>&1; cp $OUT/pmc_gemm_l2/summary.txt profiles/${RND}_gemm_l2_and_stalls.txt; head -8 profiles/${RND}_gemm_l2_and_stalls.txt | cut -c1-170
bash scripts/pmc_kernel.sh ssm_lane mamba_ssm_lane_kernel --config mamba_48m --batch 2048 | tail -14; cp $OUT/pmc_ssm_lane/summary.json profiles/${RND}_mamba_ssm_counters_lane.json
bash scripts/gpu_gemm.sh f16x2 f16x2p > profiles/${RND}_gemm_durations.txt 2>&1
python bench.py --config mamba_48m --batch 2048 --steps 64 --warmup 8 --no-cpu-baseline > profiles/${RND}_bench_mamba48m_b2048.json 2>> $OUT/bench.err; cut -c1-200 profiles/${RND}_bench_mamba48m_b2048.json; echo
python bench.py --config mamba_48m --batch 2048 --steps 32 --warmup 8 --no-cpu-baseline --mamba-compat --env-act-dim 4 > profiles/${RND}_bench_mamba48m_b2048_reference_trajectory.json 2>> $OUT/bench.err; cut -c1-200 profiles/${RND}_bench_mamba48m_b2048_reference_trajectory.json; echo
bash scripts/gpu_sweep.sh > $OUT/sweep.txt 2>/dev/null; cp $OUT/sweep.txt profiles/${RND}_config_sweep.txt; cat $OUT/sweep.txt
cp profiles/${RND}_* $OUT/profiles_out/
{ PREFILL_MODES=chunkwise python scripts/bench_prefill.py xlstm_206m 64 512 | tail -1; PREFILL_MODES=chunkwise python scripts/bench_prefill.py xlstm_16m 1024 63 | tail -1; } > profiles/${RND}_prefill_chunkwise_bench.txt 2>/dev/null; cat profiles/${RND}_prefill_chunkwise_bench.txt
python scripts/read_ceiling.py >> profiles/${RND}_prefill_chunkwise_bench.txt 2>/dev/null
cp profiles/${RND}_* $OUT/profiles_out/
bash scripts/gpu_timeline.sh x206m --config xlstm_206m --batch 512 > /dev/null 2>&1
cp $OUT/timeline_x206m.txt profiles/${RND}_step_timeline_xlstm206m_b512.txt; cp $OUT/kernel_stats_x206m.csv profiles/${RND}_kernel_stats_xlstm206m_b512.csv
head -12 profiles/${RND}_step_timeline_xlstm206m_b512.txt | cut -c1-160
cp $OUT/pytest_gpu.log profiles/${RND}_gpu_suite_final.txt 2>/dev/null
cp profiles/${RND}_* $OUT/profiles_out/
# round 6 additions: the 8-phase GEMM and the narrow-output kernels beside the kernels they stand next to; the long headline run
# (SURVEY 8d's 64 + 512 steps); the weight-distribution horizon report of the suite above
bash scripts/gpu_gemm.sh f16x2p f16x2p8 > profiles/${RND}_gemm_8phase_durations.txt 2>&1
bash scripts/gpu_gemm.sh f16x2 narrow narrow16 2>&1 | grep -E "mamba_x" > profiles/${RND}_gemm_narrow_durations.txt
python bench.py --steps 512 --warmup 64 --no-cpu-baseline --no-configs > profiles/${RND}_bench_xlstm16m_b4096_512steps.json 2>> $OUT/bench.err; cut -c1-200 profiles/${RND}_bench_xlstm16m_b4096_512steps.json; echo
cp $OUT/horizon_report.json profiles/${RND}_horizon_report.json 2>/dev/null
cp profiles/${RND}_* $OUT/profiles_out/
# prefill (C5): kernel stats and lane view of one lram_prefill; the chunk cell forms and the chunk lanes against each other
PREFILL_MODES=chunkwise bash scripts/gpu_prof_prefill.sh xlstm_206m 64 512 > $OUT/prefill_prof.txt 2>&1
f=$(find $OUT/prof_prefill_xlstm_206m -name "*kernel_stats.csv" | head -1); cp $f profiles/${RND}_kernel_stats_prefill_xlstm206m_b64_l512.csv
t=$(find $OUT/prof_prefill_xlstm_206m -name "*kernel_trace.csv" | head -1); python scripts/prefill_timeline.py $t > profiles/${RND}_prefill_timeline_xlstm206m_b64_l512.txt; rm -rf $OUT/prof_prefill_xlstm_206m
{ echo "# scripts/bench_prefill.py, same box, two rounds.  LRAM_PREFILL_CHUNK: 2 = chunk cell on the fp32-input matrix cores (rounds 1-5), three lanes;"
  echo "# 3 = bf16x3 cell, one chunk at a time (no lanes); 1 = default (bf16x3 cell, three chunks in flight); 0 = token-sequential kernels (4 timesteps per chunk, three chunks in flight as well)"
  for r in 1 2; do for v in 0 2 3 1; do
    LRAM_PREFILL_CHUNK=$v PREFILL_MODES=chunkwise python scripts/bench_prefill.py xlstm_206m 64 512 2>/dev/null | tail -1 | sed "s/^/LRAM_PREFILL_CHUNK=$v /"
    LRAM_PREFILL_CHUNK=$v PREFILL_MODES=chunkwise python scripts/bench_prefill.py xlstm_16m 64 512 2>/dev/null | tail -1 | sed "s/^/LRAM_PREFILL_CHUNK=$v /"
  done; done; } > profiles/${RND}_ab_prefill_cell_form_and_lanes.txt
cat profiles/${RND}_ab_prefill_cell_form_and_lanes.txt
cp profiles/${RND}_* $OUT/profiles_out/
