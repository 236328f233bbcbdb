#!/usr/bin/env python3
"""`oeh_proj_quant_i8` (the q/k/v GEMM with the quantisers in its epilogue) against what it replaces - the library pair GEMM +
three `oeh_quantize_heads_i8` passes - and against the exact (float64) quantisation: index agreement and launch times.  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from outeffhop_amd import ops


def timeit(fn, iters=30, rounds=8):
    res = []
    for r in range(rounds):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            res.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(res))


def main():
    torch.manual_seed(0)
    for (B, S, H, K, want) in ((16, 512, 12, 768, True), (16, 512, 12, 768, False), (32, 128, 12, 768, False), (3, 48, 2, 64, True), (5, 80, 12, 768, True)):
        E = H * 64
        M = B * S
        x = torch.randn(B, S, K, device="cuda")
        x[..., ::97] *= 30.0
        wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
        bias = torch.randn(3 * E, device="cuda") * 0.1
        alphas = [0.003, 0.0025, 0.002]
        pairs = ops.split_pairs(x.view(M, K))
        ww3 = torch.cat([wi, wi * 2.0 ** -11], dim=1).t().contiguous()
        ref64 = (x.view(M, K).double() @ wi.double().t())
        specs = []
        for n in range(3):
            v = ref64[:, n * E:(n + 1) * E] * alphas[n] + bias[n * E:(n + 1) * E].double()
            lo, hi = v.min().item() * 0.9, v.max().item() * 0.9
            sc = np.float32((hi - lo) / 255.0)
            zp = float(np.clip(np.rint(-lo / sc), 0, 255))
            specs.append(ops.FakeQuantSpec(float(sc), zp))

        def old():
            acc3 = torch.mm(pairs, ww3, out_dtype=torch.float32).view(B, S, 3 * E)
            return [ops.quantize_heads_i8(acc3[..., n * E:(n + 1) * E], specs[n], H, transpose=(n == 2), want_values=(n > 0 and want), alpha=alphas[n],
                                          bias=bias[n * E:(n + 1) * E].contiguous()) for n in range(3)]

        def new():
            return ops.proj_quant_i8(pairs, wi, bias, B, S, [(alphas[n], specs[n], n == 2, n > 0 and want) for n in range(3)], pairs=True)

        o, nw = old(), new()
        torch.cuda.synchronize()
        line = f"B={B} S={S} H={H} K={K} values={int(want)}:"
        for n in range(3):
            io, inw = (o[n][0], nw[n][0]) if (n > 0 and want) else (o[n], nw[n])
            io, inw = io.contiguous().to(torch.int32), inw.contiguous().to(torch.int32)
            v = ref64[:, n * E:(n + 1) * E] * alphas[n] + bias[n * E:(n + 1) * E].double()
            ex = (torch.clamp(torch.round(v / float(specs[n].scale)) + specs[n].zero_point, 0, 255) - 128).to(torch.int32).view(B, S, H, 64).permute(0, 2, 1, 3)
            if n == 2:
                ex = ex.permute(0, 1, 3, 2)
            ex = ex.contiguous()
            d_on = (io - inw).abs()
            line += f" [{'qkv'[n]}] old!=new {float((d_on != 0).float().mean()):.1e} (max {int(d_on.max())}) old!=exact {float((io != ex).float().mean()):.1e} new!=exact {float((inw != ex).float().mean()):.1e} (max {int((inw - ex).abs().max())})"
            if n > 0 and want:
                yo, yn = o[n][1], nw[n][1]
                same = (o[n][0].contiguous() == nw[n][0].contiguous())
                if n == 2:
                    same = same.permute(0, 1, 3, 2)
                same = same.permute(0, 2, 1, 3).reshape(B, S, E) if n == 1 else same.permute(0, 2, 1, 3).reshape(B, S, E)
                line += f" values differ where indices agree: {int(((yo != yn) & same).sum())}"
        t_old, t_new = timeit(old), timeit(new)
        print(line + f" | old {t_old:.1f} us new {t_new:.1f} us", flush=True)


if __name__ == "__main__":
    main()
