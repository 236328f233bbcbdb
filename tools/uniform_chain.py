#!/usr/bin/env python3
"""How much could ANY re-balancing of the headline launch's chains buy (key split, stream-K work items, persistent pulls)?  (GPU box)
The causal OPT-125m launch (B=16 H=12 S=512 d=64) gives its 768 workgroups chains of 2 / 4 / 6 / 8 key tiles (3 840 workgroup-tiles
in all, 3 456 once the diagonal tiles' skipped blocks are counted at half).  The SAME one-pass kernel on a NON-causal problem with
512 query rows and Sk keys gives every workgroup exactly Sk / 64 tiles - a perfectly balanced launch with no combine step at all:
Sk = 320 is the same 3 840 workgroup-tiles, Sk = 256 is 3 072: the causal launch's 3 456 lie half way.  Blocks of launches of the three problems alternate
in ONE process (medians of 8 rounds), so clocks and box are common.  If the balanced launches are not faster than the causal one,
no re-ordering or splitting of the causal launch's chains can be: the launch is bound by the sum of the per-tile work, not by its
longest chain."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from outeffhop_amd import ops

FMIN = float(np.finfo(np.float32).min)
B, H, S, D = 16, 12, 512, 64


def build(Sk, causal):
    g = torch.Generator(device="cuda").manual_seed(0)
    calls = []
    for _ in range(14):
        q = (torch.randn(B, S, H * D, device="cuda", generator=g) * D ** -0.5).half().view(B, S, H, D).permute(0, 2, 1, 3)
        k = torch.randn(B, Sk, H * D, device="cuda", generator=g).half().view(B, Sk, H, D).permute(0, 2, 1, 3)
        v = torch.randn(B, Sk, H * D, device="cuda", generator=g).half().view(B, Sk, H, D).permute(0, 2, 1, 3)
        calls.append(ops.PreparedAttn(q, k, v, causal=causal, clamp_min=causal, mask_min=FMIN))
    return calls


cases = [("causal S=512 (chains 2/4/6/8)", build(512, True)), ("balanced Sk=320 (5 tiles each, same 3840 wg-tiles)", build(320, False)),
         ("balanced Sk=256 (4 tiles each: 3072 wg-tiles)", build(256, False))]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {n: [] for n, _ in cases}
for rnd in range(11):
    for name, calls in cases:
        for c_ in calls[:4]:
            c_(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            calls[i % len(calls)](stream)
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 3:
            res[name].append(e0.elapsed_time(e1) * 1e3 / 200)
base = float(np.median(res[cases[0][0]]))
for name, _ in cases:
    m = float(np.median(res[name]))
    print(f"{name:60s} {m:7.2f} us (min {min(res[name]):.2f})   / causal {m / base:.3f}")
