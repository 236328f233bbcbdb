#!/usr/bin/env python3
"""Diagnostic (GPU box): per-wave s_memtime timeline of the one-pass attention kernel: timeline.py causal B S [off_bits] [MQ] [Sk].
Needs the stamped build: `make -C outeffhop_amd/csrc timeline` (-> outeffhop_amd/lib/timeline/liboeh_hip.so; the production
kernels carry no stamp code), loaded here through OEH_LIB.
Never quote run times from this build path: the stamps perturb the schedule; read the SHARES."""
import ctypes as C
import os
import sys
import os as _os
_os.environ.setdefault("OEH_DEBUG_HOOKS", "1")  # include/oeh_debug.h
_os.environ.setdefault("OEH_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "outeffhop_amd", "lib", "timeline", "liboeh_hip.so"))

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from outeffhop_amd import _lib, ops

lib = _lib.load()
causal = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, H, S, D = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 12, (int(sys.argv[3]) if len(sys.argv) > 3 else 512), 64
MQ = int(sys.argv[5]) if len(sys.argv) > 5 else 2   # query blocks per wave: workgroups of 64 * MQ rows
SK = int(sys.argv[6]) if len(sys.argv) > 6 else S   # keys per row (cross attention: every workgroup the same number of key tiles)
lib.oeh_debug_set_variant(256 | (int(sys.argv[4]) if len(sys.argv) > 4 else 0), MQ)
lib.oeh_debug_set_stamps.argtypes = [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B, S, H * D, device="cuda", generator=g) * D ** -0.5).half().view(B, S, H, D).permute(0, 2, 1, 3)
k = torch.randn(B, SK, H * D, device="cuda", generator=g).half().view(B, SK, H, D).permute(0, 2, 1, 3)
v = torch.randn(B, SK, H * D, device="cuda", generator=g).half().view(B, SK, H, D).permute(0, 2, 1, 3)
kw = dict(causal=bool(causal), clamp_min=bool(causal), mask_min=float(np.finfo(np.float32).min))
for _ in range(3):
    ops.attn_fwd(q, k, v, **kw)
nqt = (S + 64 * MQ - 1) // (64 * MQ)
nwg = nqt * B * H
buf = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device="cuda")
lib.oeh_debug_set_stamps(C.c_void_p(buf.data_ptr()))
ops.attn_fwd(q, k, v, **kw)
torch.cuda.synchronize()
lib.oeh_debug_set_stamps(C.c_void_p(0))
st = buf.cpu().numpy().reshape(nwg, 4, 32).astype(np.int64)
names = ["start", "prologue", "loopdone", "end"] + [f"{a}{i}" for i in range(8) for a in ("bar", "iss", "cmp")]
order = [0, 1] + list(range(4, 28)) + [2, 3]
for wg in (0, nwg // 3, nwg - 1):   # block ids: heaviest q tiles first
    w0 = st[wg, :, 0].min()
    print(f"--- workgroup {wg}: s_memtime ticks since its first wave started")
    for w in range(4):
        row = st[wg, w]
        print(f" wave {w}: " + " ".join(f"{names[i]}={int(row[i] - w0)}" for i in order if row[i]))
dur = st[:, :, 3].max(axis=1) - st[:, :, 0].min(axis=1)
print("per-WG duration ticks: min/median/max", int(dur.min()), int(np.median(dur)), int(dur.max()))
# s_memrealtime (100 MHz) at start / end of every wave: real-time picture of the launch
rs, re = st[:, :, 30].min(axis=1), st[:, :, 31].max(axis=1)
r0 = rs.min()
print(f"real time: kernel span {(re.max() - r0) * 10} ns; WG start offsets min/median/max {(rs.min() - r0) * 10}/{int(np.median(rs - r0)) * 10}/{(rs.max() - r0) * 10} ns")
clk = dur / np.maximum(re - rs, 1) / 10.0  # ticks per ns = GHz
print(f"in-kernel clock (ticks / real time) median {np.median(clk):.2f} GHz")
for qt in range(nqt):
    sel = slice((nqt - 1 - qt) * B * H, (nqt - qt) * B * H)
    print(f" q tile {qt}: median WG duration {int(np.median(dur[sel]))} ticks = {int(np.median((re - rs)[sel])) * 10} ns, "
          f"median start +{int(np.median(rs[sel] - r0)) * 10} ns, median end +{int(np.median(re[sel] - r0)) * 10} ns, last end +{int((re[sel] - r0).max()) * 10} ns")
pro = (st[:, :, 1] - st[:, :, 0]).max(axis=1)
print("prologue ticks (start -> Q landed) min/median/max:", int(pro.min()), int(np.median(pro)), int(pro.max()))

# placement: which CU each workgroup ran on (HW_REG_XCC_ID, HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13])
hw = st[:, 0, 29]
xcc = (hw >> 32) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ntile = np.minimum((SK + 63) // 64, MQ * (nqt - np.arange(nwg) // (B * H)))  if causal else np.full(nwg, (SK + 63) // 64)
per_cu = {}
for w in range(nwg):
    per_cu.setdefault(int(key[w]), []).append(w)
loads = np.array([sum(int(ntile[w]) for w in ws) for ws in per_cu.values()])
cnt = np.array([len(ws) for ws in per_cu.values()])
ends = np.array([max(int(re[w] - r0) for w in ws) for ws in per_cu.values()])
print(f"CUs used {len(per_cu)}; workgroups per CU min/max {cnt.min()}/{cnt.max()}; 64-key tiles per CU min/mean/max {loads.min()}/{loads.mean():.1f}/{loads.max()}")
for lv in sorted(set(loads.tolist())):
    sel = loads == lv
    print(f"  CUs with {lv:3d} tiles: {int(sel.sum()):3d}, last workgroup ends at median +{int(np.median(ends[sel])) * 10} ns")
print("example CU -> block ids:", list(per_cu.items())[:4])
