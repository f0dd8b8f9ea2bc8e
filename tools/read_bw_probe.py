#!/usr/bin/env python3
"""Read-only streaming bandwidth of the GPU as library reductions see it (context for the rooflines of the kernels that only
read: matrix streams).  torch.sum over 2 GiB of fp64 / fp32 and torch.dot of two 1 GiB vectors, 20 repetitions each."""
import time, torch
dev = "cuda:0"
def bench(f, nbytes, name):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt*1e6:.0f} us, {nbytes/dt/1e12:.2f} TB/s", flush=True)
n = 1 << 28
a = torch.ones(n, dtype=torch.float64, device=dev)
bench(lambda: a.sum(), n * 8, "sum fp64 2 GiB")
b = a.view(torch.float32)
bench(lambda: b.sum(), n * 8, "sum fp32 2 GiB")
h = n // 2
bench(lambda: torch.dot(a[:h], a[h:]), n * 8, "dot fp64 2 x 1 GiB")
c = torch.empty_like(a)
bench(lambda: c.copy_(a), 2 * n * 8, "copy 2 GiB (read + write)")
bench(lambda: torch.add(a, 1.0, out=c), 2 * n * 8, "add scalar 2 GiB (read + write)")
bench(lambda: c.fill_(0.0), n * 8, "fill 2 GiB (write)")
