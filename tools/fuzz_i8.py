#!/usr/bin/env python3
"""Differential fuzzing of the INT8-storage core (GPU box): random index tensors on random 8-bit grids through `ops.attn_fwd_i8`
(both products on v_mfma_i32_16x16x64_i8) against the fake-quant kernels on the DEQUANTISED fp32 values (`ops.attn_fwd(fq=...)`:
operand pairs, fp32-accurate scores) - shapes, causal / key padding / none, both softmax bases, BERT and OPT order, gate, q grids
with zero point 0 / 255 (the CQ2 variant), output dtypes, the integer output (`ctx_emit_index`).  The two must agree except
where a score or probability sat on a rounding boundary of its quantiser (exact integer products against rounded fp32 ones):
such rows differ (and single steps of the context grid anywhere); a report is printed when more than 1 % of the outputs in more than two rows and 1 % of the rows are apart.
usage: python tools/fuzz_i8.py [seconds=90] [seed=0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from outeffhop_amd import ops


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    fmin = float(np.finfo(np.float32).min)
    t0, n, bad, ties = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        n += 1
        B, H, D = int(rng.integers(1, 4)), int(rng.integers(1, 5)), 64
        S = 16 * int(rng.integers(1, 33))
        causal = bool(rng.random() < 0.5)
        use_pad = bool(rng.random() < 0.4)
        base = int(rng.integers(0, 2))
        bert = bool(rng.random() < 0.4)
        use_gate = bool(rng.random() < 0.3)
        out_dt = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 3))]
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(0, 2 ** 31)))
        zq = float(rng.choice([0.0, 255.0, float(rng.integers(100, 160))]))
        grids = [ops.QuantGrid(float(rng.uniform(0.01, 0.04)) * (0.3 if zq in (0.0, 255.0) else 1.0), zq),
                 ops.QuantGrid(float(rng.uniform(0.01, 0.04)), float(rng.integers(100, 160))),
                 ops.QuantGrid(float(rng.uniform(0.01, 0.04)), float(rng.integers(100, 160)))]
        idx = [torch.randint(0, 256, (B, S, H * D), device=dev, generator=g, dtype=torch.int32).to(torch.uint8) for _ in range(3)]
        deq = [(i_.float() - gr.zero_point) * np.float32(gr.scale) for i_, gr in zip(idx, grids)]
        heads = lambda t: t.view(B, S, H, D).permute(0, 2, 1, 3)  # noqa: E731
        q8, k8 = heads(ops.centre_indices(idx[0])), heads(ops.centre_indices(idx[1]))
        v8t = ops.centre_indices(idx[2]).view(B, S, H, D).permute(0, 2, 3, 1).contiguous()
        FQ = ops.FakeQuantSpec
        sc = float(rng.uniform(0.05, 0.3))
        step = float(rng.choice([0.01, 0.02, 0.04]))
        fq = ops.AttnFakeQuant(FQ(sc, float(rng.integers(100, 160))), FQ(1.0 / 255.0, 0.0), FQ(step, float(rng.integers(110, 146))), ctx_before_gate=not bert)
        clip = bool(rng.random() < 0.3)
        kw = dict(softmax=ops.SoftmaxSpec(base, clip, -0.025 if clip else 0.0, 1.1 if clip else 1.0), causal=causal, clamp_min=bool(causal or use_pad), mask_min=fmin)
        if bert:
            kw["scale_div"] = 8.0
        else:
            kw["scale"] = 0.125
        pad = None
        if use_pad:
            pad = torch.zeros(B, S, device=dev)
            for b in range(B):
                L = int(rng.choice([S, rng.integers(0, S + 1)]))
                if rng.random() < 0.2 and L < S:
                    pad[b, : S - L] = fmin
                else:
                    pad[b, L:] = fmin
        gate = torch.rand(B, H, S, 1, device=dev, generator=g) if use_gate else None
        desc = f"B={B} H={H} S={S} clip={int(clip)} causal={int(causal)} pad={int(use_pad)} base={base} bert={int(bert)} gate={int(use_gate)} out={str(out_dt)[6:]} zq={zq}"
        try:
            got = ops.attn_fwd_i8(q8, k8, v8t, grids, fq=fq, out_dtype=out_dt, key_pad_mask=pad, gate=gate, **kw).float()
        except Exception as e:  # noqa: BLE001
            print("ERROR", type(e).__name__, str(e)[:100], desc, flush=True)
            bad += 1
            continue
        ref = ops.attn_fwd(heads(deq[0]), heads(deq[1]), heads(deq[2]), fq=fq, key_pad_mask=pad, key_pad_boolean=True, gate=gate, **kw)
        tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 2e-2}[out_dt]
        err = (got - ref).abs()
        lim = 0.5 * step + tol + tol * ref.abs()
        off = float((err > lim).float().mean())
        rows_bad = int((err > lim).any(dim=-1).sum())   # a tie flips ONE row's probabilities: count rows, a tiny problem has few of them
        if not torch.isfinite(got).all() or (off > 1e-2 and rows_bad > max(2, 0.01 * B * H * S)):
            print(f"MISMATCH {off:.2e} of the outputs apart, max {float(err.max()) / step:.2f} steps | {desc}", flush=True)
            bad += 1
        elif off > 0:
            ties += 1
        if not use_gate or bert:  # the integer output: scale * integers reproduces the values
            import dataclasses
            rel = ops.attn_fwd_i8(q8, k8, v8t, grids, fq=dataclasses.replace(fq, ctx_emit_index=True), out_dtype=torch.float16, key_pad_mask=pad, gate=gate, **kw)
            val = ops.attn_fwd_i8(q8, k8, v8t, grids, fq=fq, out_dtype=torch.float32, key_pad_mask=pad, gate=gate, **kw)
            if not torch.equal(rel.float() * np.float32(step), val):
                print("EMIT MISMATCH", desc, flush=True)
                bad += 1
    print(f"{n} problems in {time.time() - t0:.0f} s: {bad} disagreements, {ties} problems with rounding-boundary ties")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
