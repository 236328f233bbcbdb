#!/usr/bin/env python3
"""Kernel micro-benchmark (GPU box): mean launch time of oeh_attn_fwd for a list of shapes / options,
rotating over enough buffer sets to exceed the 256 MiB Infinity Cache.  Usage:
    python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1" "B=32,H=12,S=128,D=64,pad=1" ...
keys: B H S Sk(keys per row, default S) D causal pad clip int8 dtype(f16|bf16|f32) full iters reps gate base graph(=launches per captured graph)
      gmlp (>= 0: per-token gate predictor with that many hidden units evaluated in the kernel; 0 = Linear)
      ab=<other liboeh_hip.so>: same-process A/B - blocks of launches alternate between the built library and the other one
      off (bit mask of kernel variants to disable: 2 = one-pass, 4 = full-row; 64 / 128 = their fp32-storage forms; 256 = one-pass also for Sk <= 128; 512 = no snake placement)  mq (force one-pass query blocks per wave)
"""
import ctypes as C
import sys
import os as _os
_os.environ.setdefault("OEH_DEBUG_HOOKS", "1")  # include/oeh_debug.h
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from outeffhop_amd import _lib, ops


def run(spec):
    kv = dict(B=16, H=12, S=512, Sk=0, D=64, causal=0, pad=0, clip=0, int8=0, dtype="f16", full=0, iters=300, gate=0, base=1, off=0, mq=0, reps=1, graph=0, gmlp=-1, ab="", i8=0, hg=0, padbool=1)
    for item in spec.split(","):
        k, v = item.split("=")
        kv[k] = v if k in ("dtype", "ab") else int(v)
    B, H, S, D = kv["B"], kv["H"], kv["S"], kv["D"]
    Sk = kv["Sk"] or S  # keys per row (default: self attention)
    os.environ["OEH_HEAD_GROUP"] = str(kv["hg"])  # fp32-storage kernels: block order in groups of this many heads (debug hook)
    _lib.load().oeh_debug_set_variant(kv["off"], kv["mq"])
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[kv["dtype"]]
    eb = 4 if dt == torch.float32 else 2
    per_set = 2 * B * H * (S + Sk) * D * eb
    nsets = max(2, int(700e6 // per_set) + 1)
    nsets = min(nsets, 64)
    fmin = float(np.finfo(np.float32).min)
    sets = []
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(nsets):
        q = (torch.randn(B, S, H * D, device="cuda", generator=g) * D ** -0.5).to(dt).view(B, S, H, D).permute(0, 2, 1, 3)
        k = torch.randn(B, Sk, H * D, device="cuda", generator=g).to(dt).view(B, Sk, H, D).permute(0, 2, 1, 3)
        v = torch.randn(B, Sk, H * D, device="cuda", generator=g).to(dt).view(B, Sk, H, D).permute(0, 2, 1, 3)
        sets.append((q, k, v))
    pad = None
    if kv["pad"]:
        pad = torch.zeros(B, S, device="cuda")
        lens = torch.randint(S // 2, S + 1, (B,))
        for b, n in enumerate(lens.tolist()):
            pad[b, n:] = fmin
    full = None
    if kv["full"]:
        full = torch.full((S, S), fmin, device="cuda").triu(1)[None, None].expand(B, 1, S, S).contiguous()
    gate = torch.rand(B, H, S, 1, device="cuda") if kv["gate"] else None
    gmlp = None
    if kv["gmlp"] >= 0:
        m = kv["gmlp"]
        hid = torch.randn(B, S, H * D, device="cuda", generator=g).to(dt)
        w1 = torch.randn((H, m, D) if m else (H, D), device="cuda") * 0.05
        b1 = torch.zeros((H, m) if m else (H,), device="cuda")
        gmlp = ops.GatePredictor(hid, w1, b1, torch.randn(H, m, device="cuda") if m else None, torch.zeros(H, device="cuda") if m else None)
    fq = None
    if kv["int8"]:
        FQ = ops.FakeQuantSpec
        fq = ops.AttnFakeQuant(FQ(0.08, 128.0), FQ(1.0 / 255.0, 0.0), FQ(0.02, 128.0))
    spec_sm = ops.SoftmaxSpec(kv["base"], bool(kv["clip"]), -0.025 if kv["clip"] else 0.0, 1.1 if kv["clip"] else 1.0)
    out = torch.empty(B, S, H, D, dtype=dt, device="cuda").permute(0, 2, 1, 3)
    kw = dict(softmax=spec_sm, causal=bool(kv["causal"]), clamp_min=bool(kv["causal"] or kv["full"]), key_pad_mask=pad, key_pad_boolean=pad is not None and bool(kv.get("padbool", 1)), full_mask=full,
              gate=gate, fq=fq, mask_min=fmin, out=out, gate_mlp=gmlp)
    if kv["i8"]:  # INT8 storage on the integer matrix cores (dtype = the output's): random indices, v transposed
        FQ = ops.FakeQuantSpec
        kw = dict(softmax=spec_sm, causal=bool(kv["causal"]), clamp_min=bool(kv["causal"]), gate=gate, mask_min=fmin, out_dtype=dt, scale=D ** -0.5,
                  fq=ops.AttnFakeQuant(FQ(0.08, 128.0), FQ(1.0 / 255.0, 0.0), FQ(0.02, 128.0)))
        grids = (ops.QuantGrid(0.03, 131.0), ops.QuantGrid(0.03, 120.0), ops.QuantGrid(0.03, 128.0))
        sets = []
        for _ in range(nsets):
            qi, ki, vi = (torch.randint(0, 256, (B, S, H * D), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8) for _ in range(3))
            sets.append((ops.centre_indices(qi).view(B, S, H, D).permute(0, 2, 1, 3), ops.centre_indices(ki).view(B, S, H, D).permute(0, 2, 1, 3),
                         ops.centre_indices(vi).view(B, S, H, D).permute(0, 2, 3, 1).contiguous()))
        calls = [ops.PreparedAttn(*st, i8_grids=grids, **kw) for st in sets]
    else:
        calls = [ops.PreparedAttn(*st, **kw) for st in sets]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(10):
        calls[i % nsets](stream)
    torch.cuda.synchronize()
    n = kv["iters"]
    if kv["ab"]:  # boxes differ by +-8 % and clocks wander within a run: compare two builds inside one process, interleaved
        other = C.CDLL(kv["ab"], mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND)  # DEEPBIND: its own kernels, not the already loaded library's
        other.oeh_attn_fwd.argtypes = calls[0]._fn.argtypes
        other.oeh_attn_fwd.restype = C.c_int
        fns = {"built": calls[0]._fn, "other": other.oeh_attn_fwd}
        res = {k_: [] for k_ in fns}
        for rnd in range(12):
            for name, fn in fns.items():
                for c_ in calls[:4]:
                    fn(*c_._args, stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(n):
                    if fn(*calls[i % nsets]._args, stream) != 0:
                        raise RuntimeError("launch failed")
                e1.record()
                torch.cuda.synchronize()
                if rnd >= 2:
                    res[name].append(e0.elapsed_time(e1) * 1e3 / n)
        mb, mo = float(np.median(res["built"])), float(np.median(res["other"]))
        print(f"{spec[:60]:60s} built {mb:7.2f} us (min {min(res['built']):.2f})   other {mo:7.2f} us (min {min(res['other']):.2f})   built/other {mb / mo:.4f}", flush=True)
        return
    graph = None
    if kv["graph"]:  # replay a captured graph of `graph` launches instead of launching one by one
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            cs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for i in range(kv["graph"]):
                calls[i % nsets](cs)
        n = max(1, n // kv["graph"]) * kv["graph"]
    samples = []
    for _ in range(max(1, kv["reps"])):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if graph is not None:
            for i in range(n // kv["graph"]):
                graph.replay()
        else:
            for i in range(n):
                calls[i % nsets](stream)
        e1.record()
        torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) * 1e3 / n)
    us = float(np.median(samples))  # reps > 1: median of the repeats (boxes and clocks wander by a few %)
    alg = per_set
    var = "i8mfma" if kv["i8"] else ops.attn_variant(B, H, S, Sk, D, dt, fq=bool(kv["int8"]), clip=bool(kv["clip"]))
    print(f"{spec:60s} {us:8.2f} us  {alg / us / 1e3:8.1f} GB/s alg  frac {alg / us / 1e3 / 8000:.3f}  "
          f"{4 * B * H * S * S * D / us / 1e6:7.1f} TF(dense)  sets={nsets}  [{var}]", flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)
