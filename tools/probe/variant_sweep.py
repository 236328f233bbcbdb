#!/usr/bin/env python3
"""Probe (GPU box): for a grid of shapes, the library's own kernel choice against the alternatives (one-pass forced / disabled, full-row
disabled): where is the pick not the fastest?  usage: python tools/probe/variant_sweep.py"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = int(os.environ.get("SWEEP_D", "64")); H = int(os.environ.get("SWEEP_H", "12")); DT = os.environ.get("SWEEP_DT", "f16")
specs = []
for S, B in ((64, 128), (128, 64), (192, 48), (256, 32), (320, 24), (384, 24), (512, 16), (768, 12), (1024, 8)):
    for causal in (0, 1):
        for extra in ("", ",clip=1", ",int8=1"):
            base = f"B={B},H={H},S={S},D={D},dtype={DT},causal={causal}{extra},iters=150"
            specs += [base, base + (",off=66" if DT == "f32" else ",off=2"), base + (",off=132" if DT == "f32" else ",off=4"), base + ",off=256"]
out = subprocess.run([sys.executable, os.path.join(root, "tools", "microbench.py")] + specs, capture_output=True, text=True, cwd=root).stdout
rows = [l for l in out.splitlines() if " us " in l]
for i in range(0, len(rows), 4):
    grp = rows[i:i + 4]
    vals = []
    for l in grp:
        name = l.split()[0]
        us = float(l.split(" us")[0].split()[-1])
        var = l[l.rindex("["):]
        vals.append((us, name, var))
    best = min(vals)
    flag = "  <-- pick is not the fastest" if best[0] < 0.95 * vals[0][0] else ""
    print(f"{vals[0][1]:40s} pick {vals[0][0]:7.2f} {vals[0][2]:32s} | " + "  ".join(f"{v[0]:7.2f}{v[2]}" for v in vals[1:]) + flag)
