#!/usr/bin/env python3
"""What this box's HBM delivers to simple streaming kernels (GPU box): copy (read + write), fill (write), sum (read) over 4 GiB buffers —
the practical ceiling the blend and sampler kernels' achieved GB/s are to be read against (the 8 TB/s of the data sheet is not reachable)."""
import time, torch
n = 1 << 30  # float32 elements: 4 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def timed(f, k=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
t = timed(lambda: b.copy_(a)); print("copy  4 GiB -> 4 GiB : %.3f ms  %.2f TB/s (read + write)" % (t * 1e3, 2 * 4 * n / t / 1e12))
t = timed(lambda: b.fill_(1.0)); print("fill  4 GiB          : %.3f ms  %.2f TB/s (write)" % (t * 1e3, 4 * n / t / 1e12))
t = timed(lambda: a.sum()); print("sum   4 GiB          : %.3f ms  %.2f TB/s (read)" % (t * 1e3, 4 * n / t / 1e12))
t = timed(lambda: torch.add(a, b, out=b)); print("add   2 x 4 GiB -> 4 GiB: %.3f ms  %.2f TB/s (2 reads + 1 write)" % (t * 1e3, 3 * 4 * n / t / 1e12))
