#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | cut -c1-300
bash scripts/gpu_sweep.sh 2>&1 | cut -c1-260
