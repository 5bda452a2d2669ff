"""STREAM-like ceilings of this box through the library's own kernels: lram_stream_copy (LRAM_COPY_VARIANT selects
the kernel shape) and lram_stream_rmw; HIP-event timed, 1 GiB per array."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lram_amd.engine import stream_copy, stream_rmw

n = 256 * 1024 * 1024
src = torch.rand(n, device="cuda")
dst = torch.empty(n, device="cuda")


def rate(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return reps * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


c = rate(lambda: stream_copy(dst, src))
assert torch.equal(dst, src)
print(f"variant {os.environ.get('LRAM_COPY_VARIANT', 'default')}: copy {c:.0f} GB/s   rmw {rate(lambda: stream_rmw(src)):.0f} GB/s   "
      f"torch copy_ {rate(lambda: dst.copy_(src)):.0f} GB/s")
