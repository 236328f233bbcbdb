#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wave s_memtime timeline of the K/V-resident attention kernel.
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
nwg = 192
buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device="cuda")
lib.oeh_debug_set_stamps(C.c_void_p(buf.data_ptr()))
ops.attn_fwd(q, k, v, **kw)
torch.cuda.synchronize()
lib.oeh_debug_set_stamps(C.c_void_p(0))
st = buf.cpu().numpy().reshape(nwg, 8, 16).astype(np.int64)
t0 = st[:, :, 0].min()
names = ["start", "DMA issued", "own K landed", "barrier1", "b0 QK", "b0 max", "b0 exp", "b0 inv", "barrier2", "b0 PV", "b0 st",
         "b1 QK", "b1 max", "b1 exp", "b1 inv", "b1 PV"]
print(f"all-WG kernel span: {(st.max() - t0) / 100.0:.1f} us-ish at 100 MHz? raw ticks {st.max() - t0}")
for wg in (0, 95, 191):
    print(f"--- workgroup {wg} (ticks relative to the first stamp of the launch; s_memtime ticks)")
    for w in range(8):
        row = st[wg, w]
        w0 = st[wg, :, 0].min()
        rel = [(int(x - w0) if x else -1) for x in row]
        print(f" wave {w}: " + " ".join(f"{n}={r}" for n, r in zip(names, rel) if r >= 0))
dur = st[:, :, :15].max(axis=(1, 2)) - st[:, :, 0].min(axis=1)
print("per-WG duration ticks: min/median/max", int(dur.min()), int(np.median(dur)), int(dur.max()))
print("WG start spread ticks:", int(st[:, :, 0].min(axis=1).max() - t0))
