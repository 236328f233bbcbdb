#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wave s_memtime timeline of the one-pass attention kernel (flash16, MQ=2).
Never quote run times from this build path: the stamps perturb the schedule; read the SHARES."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from outeffhop_amd import _lib, ops

B, H, S, D = 16, 12, 512, 64
causal = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = _lib.load()
lib.oeh_debug_set_stamps.argtypes = [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B, S, H * D, device="cuda", generator=g) * D ** -0.5).half().view(B, S, H, D).permute(0, 2, 1, 3)
k = torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3)
v = torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3)
kw = dict(causal=bool(causal), clamp_min=bool(causal), mask_min=float(np.finfo(np.float32).min))
for _ in range(3):
    ops.attn_fwd(q, k, v, **kw)
nwg = 768
buf = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device="cuda")
lib.oeh_debug_set_stamps(C.c_void_p(buf.data_ptr()))
ops.attn_fwd(q, k, v, **kw)
torch.cuda.synchronize()
lib.oeh_debug_set_stamps(C.c_void_p(0))
st = buf.cpu().numpy().reshape(nwg, 4, 32).astype(np.int64)
names = ["start", "prologue", "loopdone", "end"] + [f"{a}{i}" for i in range(14) for a in ("bar", "cmp")]
order = [0, 1] + list(range(4, 32)) + [2, 3]
for wg in (0, 100, 300, 500, 767):   # block ids: heaviest q tiles first
    w0 = st[wg, :, 0].min()
    print(f"--- workgroup {wg}: s_memtime ticks since its first wave started")
    for w in range(4):
        row = st[wg, w]
        print(f" wave {w}: " + " ".join(f"{names[i]}={int(row[i] - w0)}" for i in order if row[i]))
dur = st[:, :, 3].max(axis=1) - st[:, :, 0].min(axis=1)
print("per-WG duration ticks: min/median/max", int(dur.min()), int(np.median(dur)), int(dur.max()))
t0 = st[:, :, 0].min()
print("kernel span ticks (first start .. last end):", int(st[:, :, 3].max() - t0))
for qt in range(4):
    sel = slice((3 - qt) * 192, (4 - qt) * 192)
    print(f" q tile {qt}: median WG duration {int(np.median(dur[sel]))} ticks, median start {int(np.median(st[sel, :, 0].min(axis=1) - t0))}")
