#!/usr/bin/env python3
"""Launch time of oeh_proj_quant_i8 on the OPT-125m shape (OEH_GEMM_DBG=1: no epilogue, 2: two K steps only, 3: both)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from outeffhop_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proj_check import timeit
B, S, H, K = 16, 512, 12, 768
E = H * 64
x = torch.randn(B, S, K, device="cuda")
wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
bias = torch.randn(3 * E, device="cuda") * 0.1
pairs = ops.split_pairs(x.view(B * S, K))
sp = ops.FakeQuantSpec(0.05, 128.0)
for want in (True, False):
    t = timeit(lambda: ops.proj_quant_i8(pairs, wi, bias, B, S, [(0.003, sp, n == 2, n > 0 and want) for n in range(3)], pairs=True))
    print(f"dbg={os.environ.get('OEH_GEMM_DBG', '0')} values={int(want)}: {t:.1f} us", flush=True)
