#!/usr/bin/env python3
"""Where the HOST time of one quantised OPT-125m attention forward goes (the eager forward is host-bound: 154 us against 122 us of GPU
work): cProfile over 300 forwards, top functions by cumulative time.  GPU box."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from outeffhop_amd import quantization as Q
from outeffhop_amd.opt_attention import OPTAttentionWithExtras
from outeffhop_amd.softmax import SOFTMAX_MAPPING

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S, E, H = 16, 512, 768, 12
with torch.no_grad():
    org = OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
    import outeffhop_amd as oa
    cfg = oa.get_quant_config()
    cfg.act_quant.options = dict(percentile=99.999)
    qm = Q.QuantizedOPTAttentionWithExtras(org, **{**oa.val_qparams(cfg), "quant_dict": {}}).to(dev).eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    fmin = torch.finfo(torch.float32).min
    mask = torch.full((S, S), fmin, device=dev).triu(1)[None, None].expand(B, 1, S, S).contiguous()
    for _ in range(2):
        qm(torch.randn(B, S, E, device=dev), attention_mask=mask)
    qm.fix_ranges()
    x = torch.randn(B, S, E, device=dev)
    for _ in range(50):
        qm(x, attention_mask=mask)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        qm(x, attention_mask=mask)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host time per forward {1e6 * (t1 - t0) / n:.1f} us; with the final sync {1e6 * (t2 - t0) / n:.1f} us")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        qm(x, attention_mask=mask)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(22)
    st.sort_stats("tottime").print_stats(22)
