"""IMPALA-CNN front end: hand-written kernels (lram_embed_images) vs the PyTorch / MIOpen module, ms per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd import init_state_dict
from lram_amd.config import ModelSpec
from lram_amd.engine import Engine
from tests.torch_image_encoder import ImageEncoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
spec = ModelSpec(backbone="xlstm", d_model=D, n_blocks=2, slstm_at=[1])
sd = init_state_dict(spec, seed=0, with_image_encoder=True)
eng = Engine(spec, sd, B, device="cuda:0")
img = torch.randint(0, 256, (B, 3, 64, 64), dtype=torch.uint8, device="cuda:0")
mi = ImageEncoder.from_state_dict(sd, (3, 64, 64), D).cuda()
out = torch.empty(B, D, device="cuda:0")
for name, fn in (("lram_embed_images", lambda: eng.embed_images(img, out)), ("PyTorch / MIOpen", lambda: mi(img))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name}: B={B} D={D}  {dt*1e3:.3f} ms per batch  ({B/dt:,.0f} frames/s, {65e6*B/dt/1e12:.1f} TFLOP/s)")
