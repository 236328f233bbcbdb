#!/usr/bin/env python3
"""Launch times of oeh_proj_quant_i8 (OEH_GEMM_DBG bits: diagnostic; OEH_GEMM_TILE=1|2 forces the 128x288 | 64x192 tile) and of what it
replaces, on the OPT-125m and BERT-base layer shapes; the prepared calls are replayed from a HIP graph so that host time stays out."""
import os, sys
os.environ.setdefault("OEH_DEBUG_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from outeffhop_amd import ops


def gtime(fn, iters=20, rounds=8):
    """fn captured `iters` times into one graph, replayed: GPU time per call"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    res = []
    for r in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            res.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(res))


def main():
    tag = f"dbg={os.environ.get('OEH_GEMM_DBG', '0')} tile={os.environ.get('OEH_GEMM_TILE', 'auto')}"
    for (B, S, H, K) in ((16, 512, 12, 768), (32, 128, 12, 768)):
        E = H * 64
        M = B * S
        x = torch.randn(B, S, K, device="cuda")
        wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
        bias = torch.randn(3 * E, device="cuda") * 0.1
        pairs = ops.split_pairs(x.view(M, K))
        sp = ops.FakeQuantSpec(0.05, 128.0)
        ww3 = torch.cat([wi, wi * 2.0 ** -11], dim=1).t().contiguous()
        line = f"{tag} B={B} S={S}:"
        for want in (True, False):
            t = gtime(lambda: ops.proj_quant_i8(pairs, wi, bias, B, S, [(0.003, sp, n == 2, n > 0 and want) for n in range(3)], pairs=True))
            line += f" qkv values={int(want)} {t:.1f} us"
            if tag.startswith("dbg=0 tile=auto"):
                def old():
                    acc3 = torch.mm(pairs, ww3, out_dtype=torch.float32).view(B, S, 3 * E)
                    return [ops.quantize_heads_i8(acc3[..., n * E:(n + 1) * E], sp, H, transpose=(n == 2), want_values=(n > 0 and want), alpha=0.003,
                                                  bias=bias[n * E:(n + 1) * E]) for n in range(3)]
                line += f" (library GEMM + 3 passes {gtime(old):.1f})"
        # out_proj on the context quantiser's integers: (M, E) fp16 integers x (E, E)
        rel = torch.randint(-128, 128, (M, E), device="cuda").to(torch.float16)
        wo = wi[:E].contiguous()
        t = gtime(lambda: ops.proj_quant_values(rel, wo, bias[:E].contiguous(), 1e-4, sp, pairs=False))
        line += f" | out_proj {t:.1f} us"
        if tag.startswith("dbg=0 tile=auto"):
            wot = wo.t().contiguous()
            def oldo():
                acc = torch.mm(rel, wot, out_dtype=torch.float32)
                return ops.quantize_heads_i8(acc.view(1, -1, E), sp, H, want_values=True, alpha=1e-4, bias=bias[:E].contiguous(), want_indices=False)
            line += f" (library GEMM + pass {gtime(oldo):.1f})"
        rel8, wo8 = rel.to(torch.int8), wo.to(torch.int8)
        add = torch.zeros(E, dtype=torch.int32, device="cuda")
        t = gtime(lambda: ops.proj_quant_values(rel8, wo8, bias[:E].contiguous(), 1e-4, sp, pairs=False, acc_add=add))
        line += f" | out_proj int8 {t:.1f} us"
        t = gtime(lambda: ops.split_pairs(x.view(M, K)))
        line += f" | split_pairs {t:.1f} us"
        x2 = x.view(M, K)
        for want in (True, False):
            t = gtime(lambda: ops.proj_quant_i8(x2, wi, bias, B, S, [(0.003, sp, n == 2, n > 0 and want) for n in range(3)], pairs=True))
            line += f" | qkv from fp32 (split inside) values={int(want)} {t:.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
