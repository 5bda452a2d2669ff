#!/bin/bash
t() { echo "== $*"; env "$@" timeout 300 python -m pytest tests/test_gpu_lazy.py -q -m gpu -x -k "env_slices_and_206m" 2>&1 | tail -1; }
t A=1
t A=1
t LRAM_EMBED_FUSE=0
t LRAM_SLSTM_GATES_ONE=0
t LRAM_GEMM_SKINNY_NORM=0
t LRAM_GEMM_SKINNY_ROWS=0
t LRAM_SLSTM_FUSED_ROWS=0
t LRAM_GEMM_SKINNY_MIN=9
t LRAM_GEMM_SKINNY_FORM=0
