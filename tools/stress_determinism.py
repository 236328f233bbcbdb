#!/usr/bin/env python3
"""Determinism stress (GPU box): repeat one launch N times while a second stream runs another attention shape, and count
launches whose output differs bitwise from the first.  Found the missing `s_waitcnt lgkmcnt(0)` in front of the raw
s_barrier (oeh_common.h: barrier_mem): ~1 launch in 1000 wrong on padded inputs.  usage: stress_determinism.py N"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from outeffhop_amd import ops
fmin = float(np.finfo(np.float32).min)
N = int(sys.argv[1])
tot = 0
def case(name, B,H,S,D, kwf, dt=torch.float16):
    global tot
    for seed in range(2):
        torch.manual_seed(seed)
        q = (torch.randn(B,S,H*D, device="cuda")*0.3).to(dt).view(B,S,H,D).permute(0,2,1,3)
        k = torch.randn(B,S,H*D, device="cuda").to(dt).view(B,S,H,D).permute(0,2,1,3)
        v = torch.randn(B,S,H*D, device="cuda").to(dt).view(B,S,H,D).permute(0,2,1,3)
        pad = torch.zeros(B,S, device="cuda")
        for b in range(B): pad[b, int(S*(0.5+0.5*b/B)):] = fmin
        kw = kwf(pad)
        s2 = torch.cuda.Stream(); q2 = torch.randn(4,8,333,64, device="cuda").half()
        outs = [ops.attn_fwd(q,k,v, mask_min=fmin, **kw).clone() for _ in range(3)]
        ref = outs[0]
        bad = sum(0 if torch.equal(o, ref) else 1 for o in outs)
        for it in range(N):
            if it % 3 == 0:
                with torch.cuda.stream(s2):
                    ops.attn_fwd(q2,q2,q2, causal=True, clamp_min=True, mask_min=fmin)
            if not torch.equal(ops.attn_fwd(q,k,v, mask_min=fmin, **kw), ref): bad += 1
        print(f"{name:34s} seed {seed}: {bad} of {N} differ [{ops.attn_variant(B,H,S,S,D, dt, clip=bool(kw.get('softmax') and kw['softmax'].clip), fq=kw.get('fq') is not None)}]"); tot += bad
case("one-pass pad S=512", 16,12,512,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad))
case("one-pass causal+pad S=512", 16,12,512,64, lambda pad: dict(causal=True, clamp_min=True, key_pad_mask=pad))
case("full-row pad S=128 (BERT)", 32,12,128,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad))
case("full-row clip+pad S=512", 16,12,512,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad, softmax=ops.SoftmaxSpec(1, True, -0.025, 1.1)))
case("one-pass causal S=512", 16,12,512,64, lambda pad: dict(causal=True, clamp_min=True))
FQ = ops.FakeQuantSpec
int8 = ops.AttnFakeQuant(FQ(0.08, 128.0), FQ(1.0 / 255.0, 0.0), FQ(0.02, 128.0))
case("fp32 one-pass causal S=512", 16,12,512,64, lambda pad: dict(causal=True, clamp_min=True), torch.float32)
case("fp32 one-pass pad S=512", 16,12,512,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad), torch.float32)
case("fp32 one-pass pad S=128", 32,12,128,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad), torch.float32)
case("INT8 full-row causal S=512", 16,12,512,64, lambda pad: dict(causal=True, clamp_min=True, fq=int8))
case("INT8 fp32 causal S=512", 16,12,512,64, lambda pad: dict(causal=True, clamp_min=True, fq=int8), torch.float32)
# the two-pass forms of the one-pass kernel (the key stream restarts behind a barrier) and the small-shape kernel
clipsm = ops.SoftmaxSpec(1, True, -0.025, 1.1)
case("two-pass clip causal S=1024", 8,12,1024,64, lambda pad: dict(causal=True, clamp_min=True, softmax=clipsm))
case("two-pass clip S=640 fp32", 8,12,640,64, lambda pad: dict(causal=True, clamp_min=True, softmax=clipsm), torch.float32)
case("two-pass INT8 causal S=1024", 8,12,1024,64, lambda pad: dict(causal=True, clamp_min=True, fq=int8))
case("two-pass INT8 S=640 fp32", 8,12,640,64, lambda pad: dict(causal=True, clamp_min=True, fq=int8), torch.float32)
case("small-shape 28x28", 224,4,28,64, lambda pad: dict(scale=0.125), torch.float32)
# round 3: the grid chain with a key-padding vector (full-row FQ == 3), the PAD forms of the two-pass kernels, vanilla + padding on the
# one-pass kernel, a (B,1,S,S) mask on long rows
case("INT8 grid + pad S=128 (BERT)", 32,12,128,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad, key_pad_boolean=True, fq=int8))
case("INT8 grid causal + pad S=512", 16,12,512,64, lambda pad: dict(causal=True, clamp_min=True, key_pad_mask=pad, key_pad_boolean=True, fq=int8))
case("two-pass INT8 + pad S=704", 8,12,704,64, lambda pad: dict(causal=True, clamp_min=True, key_pad_mask=pad, key_pad_boolean=True, fq=int8))
case("two-pass clip + pad S=704", 8,12,704,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad, softmax=clipsm))
case("one-pass vanilla + pad S=640", 8,12,640,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad, softmax=ops.SoftmaxSpec(0)))
case("one-pass full mask S=640", 4,12,640,64, lambda pad: dict(scale_div=8.0, clamp_min=True, full_mask=(pad[:, None, None, :] + torch.zeros(pad.shape[0], 1, 640, 640, device="cuda")).contiguous()))
# round 4: the INT8-storage core (every V^T tile requested when the K phase ends, resident through the V^T phase: DMA into freed K slots
# behind one barrier, one wait + barrier for the whole phase), 16-bit output (staged through K slots 0, 1 behind a barrier on 8-tile rows)
def case_i8(name, B, H, S, out_dtype, causal, padded):
    global tot
    g = torch.Generator(device="cuda").manual_seed(1)
    qi, ki, vi = (torch.randint(0, 256, (B, S, H * 64), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8) for _ in range(3))
    qc = ops.centre_indices(qi).view(B, S, H, 64).permute(0, 2, 1, 3)
    kc = ops.centre_indices(ki).view(B, S, H, 64).permute(0, 2, 1, 3)
    vt = ops.centre_indices(vi).view(B, S, H, 64).permute(0, 2, 3, 1).contiguous()
    grids = (ops.QuantGrid(0.03, 131.0), ops.QuantGrid(0.03, 120.0), ops.QuantGrid(0.03, 128.0))
    pad = None
    if padded:
        pad = torch.zeros(B, S, device="cuda")
        for b in range(B): pad[b, int(S * (0.5 + 0.5 * b / B)):] = fmin
    kw = dict(fq=int8, out_dtype=out_dtype, scale=0.125, causal=causal, clamp_min=causal, mask_min=fmin, key_pad_mask=pad)
    s2 = torch.cuda.Stream(); q2 = torch.randn(4, 8, 333, 64, device="cuda").half()
    ref = ops.attn_fwd_i8(qc, kc, vt, grids, **kw).clone()
    bad = 0
    for it in range(N):
        if it % 3 == 0:
            with torch.cuda.stream(s2):
                ops.attn_fwd(q2, q2, q2, causal=True, clamp_min=True, mask_min=fmin)
        if not torch.equal(ops.attn_fwd_i8(qc, kc, vt, grids, **kw), ref): bad += 1
    print(f"{name:34s}        : {bad} of {N} differ [i8mfma]"); tot += bad
case_i8("int8 storage causal S=512 f32 out", 16, 12, 512, torch.float32, True, False)
case_i8("int8 storage causal S=512 f16 out", 16, 12, 512, torch.float16, True, False)
case_i8("int8 storage pad S=512 bf16 out", 16, 12, 512, torch.bfloat16, False, True)
case_i8("int8 storage pad S=128 (BERT)", 32, 12, 128, torch.float32, False, True)
case_i8("int8 storage causal S=448 f16 out", 16, 12, 448, torch.float16, True, True)
case("two-pass clip + full mask S=640", 4,12,640,64, lambda pad: dict(scale_div=8.0, clamp_min=True, softmax=clipsm, full_mask=(pad[:, None, None, :] + torch.zeros(pad.shape[0], 1, 640, 640, device="cuda")).contiguous()))
case("two-pass vanilla clip + pad S=704", 8,12,704,64, lambda pad: dict(scale_div=8.0, key_pad_mask=pad, softmax=ops.SoftmaxSpec(0, True, -0.003, 1.003)))
# round 4: the projection GEMM (LDS-DMA ring of two slots ordered by vmcnt(0) + one barrier per step; the epilogue's LDS images reuse
# the slots behind a barrier) - both tile shapes, the pair and the plain operand form
def case_proj(name, B, S, H, K, pairs, values):
    global tot
    torch.manual_seed(3)
    E, M = H * 64, B * S
    x = torch.randn(M, K, device="cuda")
    a = ops.split_pairs(x) if pairs else x.half()
    wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
    bias = torch.randn(3 * E, device="cuda") * 0.1
    sp = [ops.FakeQuantSpec(0.03, 131.0), ops.FakeQuantSpec(0.035, 124.0), ops.FakeQuantSpec(0.03, 128.0)]
    segs = [(0.003, sp[n], n == 2, n > 0 and values) for n in range(3)]
    flat = lambda r: [t for o in r for t in (o if isinstance(o, tuple) else (o,))]
    ref = [t.clone() for t in flat(ops.proj_quant_i8(a, wi, bias, B, S, segs, pairs=pairs))]
    s2 = torch.cuda.Stream(); q2 = torch.randn(4, 8, 333, 64, device="cuda").half()
    bad = 0
    for it in range(N):
        if it % 3 == 0:
            with torch.cuda.stream(s2):
                ops.attn_fwd(q2, q2, q2, causal=True, clamp_min=True, mask_min=fmin)
        got = flat(ops.proj_quant_i8(a, wi, bias, B, S, segs, pairs=pairs))
        if not all(torch.equal(g, r) for g, r in zip(got, ref)): bad += 1
    print(f"{name:34s}        : {bad} of {N} differ [oeh_gemm]"); tot += bad
case_proj("projections OPT pairs + values", 16, 512, 12, 768, True, True)
case_proj("projections BERT pairs (small tile)", 32, 128, 12, 768, True, False)
case_proj("projections fp16 activations", 16, 512, 12, 768, False, False)
case_proj("projections ragged 5 x 80", 5, 80, 12, 768, True, True)
sys.exit(1 if tot else 0)
