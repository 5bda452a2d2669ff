#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for W in 32 128; do LRAM_PERSIST_WGS=$W timeout 120 python scripts/persist_trace.py xlstm_16m 1 2>&1 | grep -v amdgpu.ids; done
