"""Read-only HBM stream ceiling (lram_stream_read) in the configuration LRAM_READ_VARIANT selects; one line per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lram_amd.engine import stream_read, stream_copy

n = 1024 * 1024 * 1024  # 4 GiB of floats: far beyond the 256 MB memory-side cache
dev = torch.device("cuda:0")
src = torch.ones(n, device=dev)
sink = torch.zeros(1024, device=dev)


def rate(fn, nbytes, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return reps * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9


r = rate(lambda: stream_read(src, sink), n * 4)
print(f"LRAM_READ_VARIANT={os.environ.get('LRAM_READ_VARIANT', '802')}: read-only {r:.0f} GB/s", flush=True)
if os.environ.get("WITH_COPY"):
    dst = torch.empty(n // 4, device=dev)
    c = rate(lambda: stream_copy(dst, src[: n // 4]), 2 * n)
    print(f"copy {c:.0f} GB/s (read + write)")
