#!/usr/bin/env python3
"""GPU: the 32x32x16 one-pass form (oeh_attn_wide.hip, debug hook bit 12) against the oracle and the production one-pass kernel."""
import os, sys
os.environ.setdefault("OEH_DEBUG_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from outeffhop_amd import _lib, ops
from oracle import oeh_oracle as O
lib = _lib.load()
fmin = float(np.finfo(np.float32).min)
ok = True
MODES = [int(a) for a in sys.argv[1:]] or [4096]
for (B, H, Sq, Sk, causal, dt, base) in [(2, 3, 512, 512, True, torch.float16, 1), (2, 2, 512, 320, False, torch.float16, 1), (1, 2, 200, 200, True, torch.float16, 0),
                                          (2, 2, 100, 300, True, torch.float16, 1), (1, 3, 384, 384, True, torch.bfloat16, 1), (1, 2, 130, 70, False, torch.float16, 1),
                                          (2, 2, 640, 640, True, torch.float16, 1)]:
    g = torch.Generator().manual_seed(Sq + Sk)
    q = (torch.randn(B, H, Sq, 64, generator=g) * 0.125).to(dt); k = torch.randn(B, H, Sk, 64, generator=g).to(dt); v = torch.randn(B, H, Sk, 64, generator=g).to(dt)
    gate = torch.rand(B, H, Sq, 1, generator=g)
    want = O.attn_core(q.float().numpy(), k.float().numpy(), v.float().numpy(), base=base, causal=causal, clamp_min=causal, gate=gate.numpy())
    kw = dict(softmax=ops.SoftmaxSpec(base), causal=causal, clamp_min=causal, mask_min=fmin, gate=gate.cuda())
    lib.oeh_debug_set_variant(0, 0)
    ref = ops.attn_fwd(q.cuda(), k.cuda(), v.cuda(), **kw).float().cpu().numpy()
    e_r = np.abs(ref - want).max()
    for mode in MODES:
        lib.oeh_debug_set_variant(mode, 0)
        name = ops.attn_variant(B, H, Sq, Sk, 64, dt, base=base, causal=causal, mask_min=fmin)
        got = ops.attn_fwd(q.cuda(), k.cuda(), v.cuda(), **kw).float().cpu().numpy()
        lib.oeh_debug_set_variant(0, 0)
        e_w = np.abs(got - want).max()
        lim = 2e-3 if dt == torch.float16 else 2e-2
        good = np.isfinite(got).all() and e_w <= lim
        ok &= bool(good)
        print(f"{name:22s} mode {mode:6d} B={B} H={H} Sq={Sq} Sk={Sk} causal={causal} {str(dt)[6:]} base={base}: err {e_w:.2e}  production err {e_r:.2e}  {'ok' if good else 'FAIL'}", flush=True)
print("ALL OK" if ok else "FAILURES")
