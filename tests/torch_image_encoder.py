"""TEST INFRASTRUCTURE (not part of the product package): IMPALA-style image observation encoder in PyTorch-ROCm / MIOpen,
an independent cross-check of the engine's own kernels (`lram_embed_images`, csrc/impala_cnn.hip), which are the only
image backend `RecurrentAgent` has.  Used by tests/ and scripts/bench_image_encoder.py.


Front end of the hot path for image domains (Atari / Procgen / Mimicgen-vision): uint8 [B,3,64,64] ->
x/255 -> 3 x (conv3x3 -> maxpool(3,2,1) -> 2 residual blocks) with 16/32/32 channels -> ReLU -> flatten ->
Linear -> ReLU.  Behaviour and state-dict key names follow the reference's `embed_image` module
(src/algos/models/image_encoders.py:10-131, built at multi_domain_discrete_dt_model.py:43-46; the /255 is
online_decision_transformer_model.py:523-525), so `embed_image.*` checkpoint entries load unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Residual(nn.Module):
    def __init__(self, depth: int):
        super().__init__()
        self.conv_0 = nn.Conv2d(depth, depth, 3, padding=1)
        self.conv_1 = nn.Conv2d(depth, depth, 3, padding=1)

    def forward(self, x):
        y = self.conv_0(F.relu(x))
        y = self.conv_1(F.relu(y))
        return x + y


class _Stage(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1)
        self.residual_0 = _Residual(cout)
        self.residual_1 = _Residual(cout)

    def forward(self, x):
        x = F.max_pool2d(self.conv(x), 3, 2, padding=1)
        return self.residual_1(self.residual_0(x))


class ImageEncoder(nn.Module):
    def __init__(self, image_shape=(3, 64, 64), features_dim: int = 512, channels=(16, 32, 32)):
        super().__init__()
        cin, hw = image_shape[0], image_shape[1]
        stages = []
        for cout in channels:
            stages.append(_Stage(cin, cout))
            cin = cout
            hw = (hw + 2 - 3) // 2 + 1
        self.cnn = nn.ModuleList(stages)
        self.linear = nn.Sequential(nn.Linear(cin * hw * hw, features_dim), nn.ReLU())

    @torch.no_grad()
    def forward(self, obs_uint8: torch.Tensor) -> torch.Tensor:
        x = obs_uint8.float() / 255.0
        for st in self.cnn:
            x = st(x)
        return self.linear(F.relu(x).flatten(1))

    @classmethod
    def from_state_dict(cls, sd, image_shape, features_dim, prefix="embed_image."):
        enc = cls(image_shape, features_dim)
        sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        enc.load_state_dict(sub, strict=True)
        return enc.eval()
