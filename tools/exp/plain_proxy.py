#!/usr/bin/env python3
"""Proxy for an fp16-output form of the projection GEMM: the plain (fp16 activations) loop with the index-only quantiser epilogue against
the library's fp16 Linear (hipBLASLt, bias inside), OPT-125m / BERT-base layer shapes.  GPU box."""
import os, sys
os.environ.setdefault("OEH_DEBUG_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from outeffhop_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proj_time import gtime
sp = ops.FakeQuantSpec(0.05, 128.0)
for (B, S, N) in ((16, 512, 2304), (16, 512, 768), (32, 128, 2304), (32, 128, 768)):
    K, M = 768, B * S
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.03).half()
    wi = torch.randint(-128, 128, (N, K), device="cuda").half()
    b = torch.randn(N, device="cuda")
    bh = b.half()
    nseg = 3 if N == 2304 else 1
    t_lib = gtime(lambda: torch.nn.functional.linear(x, w, bh))
    t_new = gtime(lambda: ops.proj_quant_i8(x, wi, b, B, S, [(0.003, sp, False, False)] * nseg, pairs=False))
    print(f"M={M} N={N}: library fp16 Linear {t_lib:.1f} us | plain loop + index epilogue {t_new:.1f} us", flush=True)
