#!/usr/bin/env python3
"""Row kernels (GPU box): oeh_softmax_rows / oeh_fake_quant / oeh_minmax on the OPT score tensor (B*H*S rows of S) against
their streaming bytes.  usage: python tools/rows_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from outeffhop_amd import ops


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for dt in (torch.float32, torch.float16):
    rows, cols = 16 * 12 * 512, 512
    xs = [torch.randn(rows, cols, device="cuda").to(dt) for _ in range(3)]
    ys = [torch.empty_like(x) for x in xs]
    eb = xs[0].element_size()
    nb = rows * cols * eb
    i = [0]

    def nxt():
        i[0] = (i[0] + 1) % 3
        return i[0]

    t = timeit(lambda: ops.softmax_rows(xs[nxt()], ops.SoftmaxSpec(1, False, 0.0, 1.0), out=ys[i[0]]))
    print(f"{str(dt):14s} softmax1 rows      {t:8.1f} us  {2 * nb / t / 1e3:7.1f} GB/s")
    t = timeit(lambda: ops.softmax_rows(xs[nxt()], ops.SoftmaxSpec(1, True, -0.025, 1.1), out=ys[i[0]]))
    print(f"{str(dt):14s} clipped softmax1   {t:8.1f} us  {2 * nb / t / 1e3:7.1f} GB/s")
    t = timeit(lambda: ops.fake_quant(xs[nxt()], ops.FakeQuantSpec(0.05, 128.0)))
    print(f"{str(dt):14s} fake_quant         {t:8.1f} us  {2 * nb / t / 1e3:7.1f} GB/s")
    t = timeit(lambda: ops.minmax(xs[nxt()]))
    print(f"{str(dt):14s} minmax             {t:8.1f} us  {nb / t / 1e3:7.1f} GB/s")
    t = timeit(lambda: ys[nxt()].copy_(xs[i[0]]))
    print(f"{str(dt):14s} (torch copy)       {t:8.1f} us  {2 * nb / t / 1e3:7.1f} GB/s")
