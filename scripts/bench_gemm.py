"""GEMM micro-benchmark: the projection shapes of the BASELINE configs through lram_gemm_bf16x3 / lram_gemm_f32.

Run under `rocprofv3 --kernel-trace` (the C-ABI test wrappers allocate and synchronise per call, so wall-clock
timing of a call is meaningless); scripts/parse_gemm_trace.py matches kernel durations to shapes by call order.
"""
import json
import sys

import torch

from lram_amd.engine import gemm_f32

SHAPES = [  # (tag, M, N, K)
    ("16m_up", 6144, 2048, 512), ("16m_down", 6144, 512, 1024), ("16m_head", 2048, 2192, 512),
    ("16m_ffn_up", 6144, 1408, 512), ("16m_ffn_down", 6144, 512, 704),
    ("mamba_in", 6144, 3072, 768), ("mamba_out", 6144, 768, 1536), ("mamba_x", 6144, 80, 1536),
    ("mamba_dt", 6144, 1536, 48),
    ("206m_up", 1536, 5120, 1280), ("206m_down", 1536, 1280, 2560),
    ("16m_up_b4096", 12288, 2048, 512), ("prefill_up", 12288 * 2, 2048, 512),
    # Mamba-48M at 2048 envs runs two slices of 1024 envs: 3072 rows per projection launch
    ("mamba_in_s", 3072, 3072, 768), ("mamba_out_s", 3072, 768, 1536), ("mamba_x_s", 3072, 80, 1536),
    ("mamba_dt_s", 3072, 1536, 48),
    # 16M at 1024 env slots: two slices of 512 envs, 1536 rows per projection launch
    ("c2_down", 1536, 512, 1024), ("c2_up_x", 1536, 1024, 512), ("c2_up", 1536, 2048, 512),
    # 206M at 512 env slots: two slices of 256 envs, 768 rows per projection launch (weights beyond one XCD's L2)
    ("206m_up_s", 768, 5120, 1280), ("206m_down_s", 768, 1280, 2560),
    # C5 prefill: 206M, 512 slots in two slices x 63-token chunks = 16128 rows per projection launch
    ("c5_up", 16128, 5120, 1280), ("c5_down", 16128, 1280, 2560),
]
REPS = 5

if __name__ == "__main__":
    kernels = sys.argv[1:] or ["bf16x3"]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    order = []
    for tag, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.05
        for kern in kernels:
            if kern in ("f16x2p", "f16x2p8") and K % 32 != 0:
                continue
            if kern in ("narrow", "narrow16") and (N > 96 or K % 64 != 0 or K < 256):
                continue
            for _ in range(REPS):
                gemm_f32(a, w, kernel=kern)
            order.append({"tag": tag, "kernel": kern, "m": M, "n": N, "k": K, "reps": REPS})
    torch.cuda.synchronize()
    print(json.dumps(order))
