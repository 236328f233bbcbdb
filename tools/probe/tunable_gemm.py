#!/usr/bin/env python3
"""Probe (GPU box): the projections' library GEMMs with and without PyTorch's TunableOp (rocBLAS / hipBLASLt solution search).
usage: python tools/probe/tunable_gemm.py"""
import os
import sys
import time
import torch


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.2:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


shapes = [("opt qkv fp16", 8192, 768, 2304, torch.float16, True), ("opt out fp16", 8192, 768, 768, torch.float16, True),
          ("opt qkv pairs", 8192, 1536, 2304, torch.float16, False), ("opt out pairs", 8192, 1536, 768, torch.float16, False),
          ("bert qkv fp16", 4096, 768, 2304, torch.float16, True), ("opt qkv triples", 8192, 2312, 2304, torch.float16, False)]
res = {}
for tuned in (False, True):
    if tuned:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(30)
        torch.cuda.tunable.set_filename("/tmp/tunable.csv")
    for name, M, K, N, dt, lin in shapes:
        x = torch.randn(M, K, device="cuda", dtype=dt)
        if lin:
            w = torch.randn(N, K, device="cuda", dtype=dt)
            b = torch.randn(N, device="cuda", dtype=dt)
            fn = lambda: torch.nn.functional.linear(x, w, b)
        else:
            w = torch.randn(K, N, device="cuda", dtype=dt)
            fn = lambda: torch.mm(x, w, out_dtype=torch.float32)
        us = timeit(fn)
        res[(name, tuned)] = us
        print(f"{name:18s} tuned={tuned}: {us:7.2f} us  {2*M*K*N/us/1e6:7.1f} TFLOP/s", flush=True)
for name, *_ in shapes:
    print(f"{name:18s} tuned/default {res[(name, True)] / res[(name, False)]:.3f}")
