#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU at the attention launches' transfer sizes (GPU box): device-to-device
copies of N/2 bytes (N bytes of HBM traffic) over rotating buffers larger than the 256 MiB Infinity Cache, timed like the
attention micro-benchmark.  Context for roofline.frac: the 8 TB/s peak is not reachable by any kernel of 50 MB."""
import torch

def run(traffic_mb):
    n = int(traffic_mb * 1e6 / 2)
    nsets = max(2, int(700e6 // (2 * n)) + 1)
    src = [torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255) for _ in range(nsets)]
    dst = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(nsets)]
    for i in range(20):
        dst[i % nsets].copy_(src[i % nsets])
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300):
            dst[i % nsets].copy_(src[i % nsets])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 300
        best = us if best is None else min(best, us)
    print(f"copy with {traffic_mb:6.1f} MB of traffic: {best:7.2f} us  {traffic_mb * 1e3 / best:7.1f} GB/s  ({traffic_mb * 1e3 / best / 8000:.3f} of 8 TB/s)")

for mb in (25.2, 50.3, 100.7, 201.3, 805.3):
    run(mb)
