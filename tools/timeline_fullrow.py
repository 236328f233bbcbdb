#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wave phase timeline of the FULL-ROW kernel (oeh_attn_fast_kernel: clipped softmax / the fused INT8 chain):
    timeline_fullrow.py [int8|clip] [B] [S]
Needs the stamped build (`make -C outeffhop_amd/csrc timeline`), loaded through OEH_LIB.  Read the SHARES, not the run times (the stamps perturb)."""
import ctypes as C
import os
import sys
os.environ.setdefault("OEH_DEBUG_HOOKS", "1")
os.environ.setdefault("OEH_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "outeffhop_amd", "lib", "timeline", "liboeh_hip.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from outeffhop_amd import _lib, ops

lib = _lib.load()
kind = sys.argv[1] if len(sys.argv) > 1 else "int8"
B, H, S, D = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 12, (int(sys.argv[3]) if len(sys.argv) > 3 else 512), 64
lib.oeh_debug_set_stamps.argtypes = [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B, S, H * D, device="cuda", generator=g) * D ** -0.5).half().view(B, S, H, D).permute(0, 2, 1, 3)
k = torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3)
v = torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3)
kw = dict(causal=True, clamp_min=True, mask_min=float(np.finfo(np.float32).min))
if kind == "int8":
    FQ = ops.FakeQuantSpec
    kw["fq"] = ops.AttnFakeQuant(FQ(0.08, 128.0), FQ(1.0 / 255.0, 0.0), FQ(0.02, 128.0))
else:
    kw["softmax"] = ops.SoftmaxSpec(1, True, -0.025, 1.1)
print("variant:", ops.attn_variant(B, H, S, S, D, torch.float16, fq=kind == "int8", clip=kind != "int8", causal=True))
for _ in range(3):
    ops.attn_fwd(q, k, v, **kw)
nqt = (S + 63) // 64
nwg = nqt * B * H
buf = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device="cuda")
lib.oeh_debug_set_stamps(C.c_void_p(buf.data_ptr()))
ops.attn_fwd(q, k, v, **kw)
torch.cuda.synchronize()
lib.oeh_debug_set_stamps(C.c_void_p(0))
st = buf.cpu().numpy().reshape(nwg, 4, 32).astype(np.int64)
rs, re = st[:, :, 30].min(axis=1), st[:, :, 31].max(axis=1)
r0 = rs.min()
print(f"real time: kernel span {(re.max() - r0) * 10} ns; WG start offsets min/median/max {(rs.min() - r0) * 10}/{int(np.median(rs - r0)) * 10}/{(rs.max() - r0) * 10} ns")
dur = st[:, :, 22].max(axis=1) - st[:, :, 0].min(axis=1)
clk = dur / np.maximum(re - rs, 1) / 10.0
print(f"in-kernel clock (ticks / real time) median {np.median(clk):.2f} GHz")
# phases per wave: prologue 0->1, K phase 1->10, chain 10->12, V phase 12->21, epilogue 21->22 (ticks)
ph = {"prologue": (0, 1), "K phase": (1, 10), "chain": (10, 12), "V phase": (12, 21), "epilogue": (21, 22), "whole": (0, 22)}
for qt in range(nqt - 1, -1, -1):
    sel = slice((nqt - 1 - qt) * B * H, (nqt - qt) * B * H)   # block ids: heaviest q tiles first
    w = st[sel].reshape(-1, 32)
    parts = "  ".join(f"{n} {int(np.median(w[:, b_] - w[:, a_]))}" for n, (a_, b_) in ph.items())
    print(f" q tile {qt} ({qt + 1} key tiles): median ticks per wave: {parts};  WG start +{int(np.median(rs[sel] - r0)) * 10} ns, end +{int(np.median(re[sel] - r0)) * 10} ns (last +{int((re[sel] - r0).max()) * 10})")
wg = 0
print("--- workgroup 0 (heaviest), wave 0: K-step barriers", [int(st[wg, 0, 2 + t] - st[wg, 0, 0]) for t in range(8) if st[wg, 0, 2 + t]], " V-step barriers", [int(st[wg, 0, 13 + t] - st[wg, 0, 0]) for t in range(8) if st[wg, 0, 13 + t]])
