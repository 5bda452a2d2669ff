cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --no-cpu-baseline --no-stream-ceilings --host-io-steps 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']), round(d['ms_per_step'],3))"; }
run --config xlstm_16m --batch 1 --steps 400 --warmup 40
run --config xlstm_16m --batch 1 --steps 400 --warmup 40 --graph
LRAM_GEMM=bf16x3 run --config xlstm_16m --batch 1 --steps 400 --warmup 40 --graph
run --config xlstm_16m --batch 32 --steps 100 --warmup 10
run --config xlstm_c1 --batch 32 --steps 200 --warmup 20
run --config xlstm_16m --batch 256 --steps 50 --warmup 10
LRAM_F16_MIN_ROWS=9 run --config xlstm_16m --batch 256 --steps 50 --warmup 10
run --config xlstm_16m --batch 512 --steps 50 --warmup 10
LRAM_F16_MIN_ROWS=100000 run --config xlstm_16m --batch 512 --steps 50 --warmup 10
run --config mamba_48m --batch 1 --steps 100 --warmup 10 --graph
run --config mamba_48m --batch 1 --steps 100 --warmup 10
