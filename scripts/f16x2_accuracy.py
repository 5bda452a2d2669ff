#!/usr/bin/env python3
"""Measured accuracy of the split-operand projection kernels against fp64, beside the exact fp32 MFMA kernel (a k-ordered
fp32 fma chain) on the same data: the table behind the bar in tests/test_gpu_parity.py::test_gemm_f16x2_matches_fp64.

    python scripts/f16x2_accuracy.py > profiles/r04_f16x2_accuracy.txt

Error metric = max over outputs of |out - ref64| / sum_k |a||w| (the condition-free scale of a dot product)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lram_amd import build  # noqa: E402
from lram_amd.engine import gemm_f32  # noqa: E402

build.build(force=False, verbose=False)
SHAPES = [(128, 128, 32), (96, 2192, 512), (300, 80, 1536), (1000, 1408, 512), (4096, 512, 1024), (6144, 3072, 768),
          (6144, 2048, 512), (6144, 512, 1024), (65, 8, 8), (257, 129, 48)]
print(f"{'m':>5} {'n':>5} {'k':>5} {'spread':>9} {'e_f32':>10} {'e_f16x2':>10} {'e_bf16x3':>10} {'f16x2/f32':>10} {'bf16x3/f32':>10}")
worst = 0.0
for m, n, k in SHAPES:
    for spread in ("rows", "elements", "tiny", "activations"):
        g = torch.Generator().manual_seed(m * 13 + n)
        a = torch.randn(m, k, generator=g) * torch.exp(torch.randn(m, 1, generator=g))
        w = torch.randn(n, k, generator=g) * torch.exp(torch.randn(n, 1, generator=g) * 0.5)
        if spread == "elements":
            a = a * torch.exp(torch.randn(m, k, generator=g) * 3.0)
            w = w * torch.exp(torch.randn(n, k, generator=g) * 2.0)
        if spread == "tiny":
            a, w = a * 1e-9, w * 1e-7
        if spread == "activations":   # what the engine feeds it: normalised rows against N(0, 0.02)-like weights
            a = torch.nn.functional.layer_norm(torch.randn(m, k, generator=g), (k,))
            w = torch.randn(n, k, generator=g) * 0.02
        ref = a.double() @ w.double().t()
        scale = a.double().abs() @ w.double().abs().t() + 1e-300
        outs = {}
        for kern in ("f32", "f16x2", "bf16x3"):
            try:
                outs[kern] = ((gemm_f32(a.cuda(), w.cuda(), None, kernel=kern).cpu().double() - ref).abs() / scale).max().item()
            except Exception:
                outs[kern] = float("nan")
        r2, r3 = outs["f16x2"] / outs["f32"], outs["bf16x3"] / outs["f32"]
        worst = max(worst, r2) if r2 == r2 else worst
        print(f"{m:5d} {n:5d} {k:5d} {spread:>9} {outs['f32']:10.3e} {outs['f16x2']:10.3e} {outs['bf16x3']:10.3e} {r2:10.2f} {r3:10.2f}")
print(f"worst f16x2 / f32 ratio: {worst:.2f}")
