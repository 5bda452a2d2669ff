#!/bin/bash
# round 4, first call: steady-state PMC traffic of the state pass + whole step, headline line, full GPU suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; RND=r04; mkdir -p $OUT; cd $R
python bench.py --steps 64 --warmup 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench.json
rm -rf $OUT/pmc; PMC_ROUND=$RND bash scripts/pmc_pass.sh 2>&1 | grep -E "rc=|hbm_bytes_per_env|state_mode|parse_pmc"
python scripts/parse_pmc_step.py $OUT/pmc $(python -c "import json; print(json.load(open('$OUT/bench.json'))['ms_per_step'])") > profiles/${RND}_whole_step_hbm_traffic.json
head -12 profiles/${RND}_whole_step_hbm_traffic.json; tail -8 profiles/${RND}_whole_step_hbm_traffic.json
mkdir -p $OUT/profiles_out; cp profiles/${RND}_* $OUT/profiles_out/
if [ "${TESTS:-1}" = "1" ]; then
  timeout 2400 python -m pytest tests -q -m gpu -x --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -12 $OUT/pytest_gpu.log | cut -c1-200
fi
