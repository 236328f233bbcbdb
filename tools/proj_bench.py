#!/usr/bin/env python3
"""The projections either side of the INT8-storage attention core on the layer shapes of BASELINE.json's configurations:
`oeh_proj_quant_i8` (one GEMM with the quantisers in its epilogue) against what it replaces, the library GEMM (hipBLASLt through
torch.mm) + one `oeh_quantize_heads_i8` pass per projection, in ONE process; one JSON line per workload, in bench.py's vocabulary.
    python tools/proj_bench.py [--steps 20] [--layers 12] [workload ...]         (GPU box)
A step = the projection launches of `layers` layers on distinct buffers (inputs resident in HBM), timed with HIP events on the launch
stream around a captured HIP graph of the step (so that host time stays out: these launches are 20-75 us).  roofline: bound mfma;
`achieved` = algorithmic flops (2 M N K: ONE fp32-grade product per term - the operand-pair form executes two fp16 MFMA products per
term, reported as mfma_flops_executed) / the kernel's mean duration; peak = 2500 TFLOP/s (fp16 dense, MI355X_MICROARCH.md; 5000 for the int8 x int8 out_proj)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from outeffhop_amd import ops

WORKLOADS = {
    "opt_qkv": dict(B=16, S=512, H=12, K=768, kind="qkv", values=True,
                    desc="OPT-125m layer B=16 S=512 E=768 fp32: q/k/v QuantLinear projections from the fp32 activations (operand pairs formed inside) -> int8 indices (v transposed) + (k, v) cache values"),
    "opt_qkv_novalues": dict(B=16, S=512, H=12, K=768, kind="qkv", values=False,
                             desc="OPT-125m layer B=16 S=512 E=768 fp32: q/k/v QuantLinear projections from the fp32 activations (operand pairs formed inside) -> int8 indices (v transposed)"),
    "opt_out_proj": dict(B=16, S=512, H=12, K=768, kind="out", values=True,
                         desc="OPT-125m layer B=16 S=512 E=768: out_proj QuantLinear on the context quantiser's int8 centred indices (v_mfma_i32_16x16x64_i8) -> fake-quantised fp32 values"),
    "bert_qkv": dict(B=32, S=128, H=12, K=768, kind="qkv", values=False,
                     desc="BERT-base layer B=32 S=128 E=768 fp32: query/key/value QuantLinear projections from the fp32 activations (operand pairs formed inside) -> int8 indices (v transposed)"),
}


def graph_time(fns, steps, warmup):
    """one step = every fn once; captured, replayed: us per step"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            for f in fns:
                f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    for _ in range(warmup):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--no-baseline", action="store_true")
    ap.add_argument("workloads", nargs="*", default=sorted(WORKLOADS))
    a = ap.parse_args()
    torch.manual_seed(0)
    for name in a.workloads:
        w = WORKLOADS[name]
        B, S, H, K, L = w["B"], w["S"], w["H"], w["K"], a.layers
        E, M = H * 64, B * S
        sp = [ops.FakeQuantSpec(0.03, 131.0), ops.FakeQuantSpec(0.035, 124.0), ops.FakeQuantSpec(0.03, 128.0)]
        new, old = [], []
        for _ in range(L):
            bias = (torch.randn(3 * E, device="cuda") * 0.1)
            if w["kind"] == "qkv":
                x = torch.randn(M, K, device="cuda")
                wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
                ww3 = torch.cat([wi, wi * 2.0 ** -11], dim=1).t().contiguous()
                segs = [(0.003, sp[n], n == 2, n > 0 and w["values"]) for n in range(3)]
                new.append(lambda x=x, wi=wi, bias=bias, segs=segs: ops.proj_quant_i8(x, wi, bias, B, S, segs, pairs=True))  # fp32 in: split inside

                def lib(x=x, ww3=ww3, bias=bias):
                    pairs = ops.split_pairs(x)
                    acc3 = torch.mm(pairs, ww3, out_dtype=torch.float32).view(B, S, 3 * E)
                    return [ops.quantize_heads_i8(acc3[..., n * E:(n + 1) * E], sp[n], H, transpose=(n == 2), want_values=(n > 0 and w["values"]), alpha=0.003,
                                                  bias=bias[n * E:(n + 1) * E]) for n in range(3)]
                old.append(lib)
                N, mult = 3 * E, 2
            else:
                c8 = torch.randint(-128, 128, (M, E), device="cuda", dtype=torch.int8)       # centred indices idx - 128 (zero point 121)
                rel = (c8.to(torch.float16) + 7.0)                                             # idx - zp, what the library path multiplies
                wo8 = torch.randint(-128, 128, (E, K), device="cuda", dtype=torch.int8)
                wo = wo8.to(torch.float16)
                wot = wo.t().contiguous()
                b1 = bias[:E].contiguous()
                add = (7 * wo8.to(torch.int64).sum(dim=1)).to(torch.int32).contiguous()
                new.append(lambda c8=c8, wo8=wo8, b1=b1, add=add: ops.proj_quant_values(c8, wo8, b1, 1e-4, sp[2], pairs=False, acc_add=add))

                def lib(rel=rel, wot=wot, b1=b1):
                    acc = torch.mm(rel, wot, out_dtype=torch.float32)
                    return ops.quantize_heads_i8(acc.view(1, -1, E), sp[2], H, want_values=True, alpha=1e-4, bias=b1, want_indices=False)
                old.append(lib)
                N, mult = E, 1
        t_new = graph_time(new, a.steps, a.warmup)
        t_old = None if a.no_baseline else graph_time(old, a.steps, a.warmup)
        flops = 2.0 * M * N * K
        kern_us = t_new / L
        line = {
            "metric": "projection_tokens_per_s", "value": B * S * L / (t_new * 1e-6), "unit": "tokens/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": t_new * 1e-3, "higher_is_better": True, "dtype": "f16 operand pairs, f32 accumulate" if mult == 2 else "int8 x int8, int32 accumulate",
            "data": "synthetic", "config": {"workload": w["desc"], "layers_per_step": L, "M": M, "N": N, "K": K},
            "kernel_us": kern_us,
            "roofline": {"bound": "mfma", "achieved": flops / kern_us * 1e-6, "peak": 2500.0 if mult == 2 else 5000.0, "unit": "TFLOP/s", "frac": flops / kern_us * 1e-6 / (2500.0 if mult == 2 else 5000.0),
                         "mfma_flops_executed": mult * flops, "executed_frac": mult * flops / kern_us * 1e-6 / (2500.0 if mult == 2 else 5000.0), "traffic": None},
            "replaces": None if t_old is None else {"what": ("oeh_split_pairs + " if w["kind"] == "qkv" else "") + "library GEMM (torch.mm, hipBLASLt) + one oeh_quantize_heads_i8 pass per projection, same process",
                                                    "us_per_layer": t_old / L, "speedup": t_old / t_new},
        }
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
