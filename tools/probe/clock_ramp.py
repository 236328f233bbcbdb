#!/usr/bin/env python3
"""GPU box: per-launch time of the headline kernel as a function of time since the process' first launch
(how long does the device take to reach its sustained clocks?).  python tools/probe/clock_ramp.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from outeffhop_amd import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
B, H, S, D = 16, 12, 512, 64
g = torch.Generator(device="cuda").manual_seed(0)
sets = []
for _ in range(12):
    q, k, v = (torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3) for _ in range(3))
    sets.append((q, k, v, torch.empty(B, S, H, D, device="cuda", dtype=torch.float16).permute(0, 2, 1, 3)))
torch.cuda.synchronize()
t0 = time.perf_counter()
rows = []
while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        for q, k, v, o in sets:
            ops.attn_fwd(q, k, v, causal=True, clamp_min=True, out=o)
    e1.record()
    torch.cuda.synchronize()
    rows.append((time.perf_counter() - t0, e0.elapsed_time(e1) * 1e3 / 240))
for i, (t, us) in enumerate(rows):
    if i < 10 or i % max(1, len(rows) // 40) == 0:
        print(f"t = {t * 1e3:8.1f} ms   {us:6.2f} us / launch")
