#!/usr/bin/env python3
"""Every size / shape threshold of `pick_variant` (outeffhop_amd/csrc/oeh_api.hip) timed on BOTH of its sides, at the boundary and one
step either side, in ONE process (GPU box): blocks of launches alternate between the two kernels (the diagnostic hook
`oeh_debug_set_variant` forces the other side), medians of 8 rounds - the only comparison that survives the +-8 % between boxes.
    python tools/dispatch_ab.py > gpurun_out/dispatch_ab.txt        (committed as profiles/rNN_dispatch_ab.txt)
A line: rule | shape | the library's pick: kernel, us | the other side: kernel, us | other / pick | verdict.
`verdict`: "pick ok" (the pick is faster or within 3 %), "FLIP" (the other side is more than 3 % faster: the rule is wrong here).
A rule whose two sides are within 3 % of each other on EVERY shape around its boundary decides nothing and can go."""
import ctypes as C
import os
import sys

os.environ.setdefault("OEH_DEBUG_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from outeffhop_amd import _lib, ops

lib = _lib.load()
FMIN = float(np.finfo(np.float32).min)
DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def build(B, H, S, D, dt, causal, clip, int8, pad, Sk=None):
    Sk = Sk or S
    eb = 4 if dt == "f32" else 2
    per_set = 2 * B * H * (S + Sk) * D * eb
    nsets = min(48, max(2, int(600e6 // per_set) + 1))
    g = torch.Generator(device="cuda").manual_seed(0)
    calls = []
    padm = None
    if pad:
        padm = torch.zeros(B, Sk, device="cuda")
        for b, n in enumerate(torch.randint(Sk // 2, Sk + 1, (B,)).tolist()):
            padm[b, n:] = FMIN
    fq = None
    if int8:
        FQ = ops.FakeQuantSpec
        fq = ops.AttnFakeQuant(FQ(0.08, 128.0), FQ(1.0 / 255.0, 0.0), FQ(0.02, 128.0))
    sm = ops.SoftmaxSpec(1, bool(clip), -0.025 if clip else 0.0, 1.1 if clip else 1.0)
    for _ in range(nsets):
        q = (torch.randn(B, S, H * D, device="cuda", generator=g) * D ** -0.5).to(DT[dt]).view(B, S, H, D).permute(0, 2, 1, 3)
        k = torch.randn(B, Sk, H * D, device="cuda", generator=g).to(DT[dt]).view(B, Sk, H, D).permute(0, 2, 1, 3)
        v = torch.randn(B, Sk, H * D, device="cuda", generator=g).to(DT[dt]).view(B, Sk, H, D).permute(0, 2, 1, 3)
        calls.append(ops.PreparedAttn(q, k, v, softmax=sm, causal=bool(causal), clamp_min=bool(causal), key_pad_mask=padm,
                                      key_pad_boolean=padm is not None, fq=fq, mask_min=FMIN))
    name = lambda: ops.attn_variant(B, H, S, Sk, D, DT[dt], fq=bool(int8), clip=bool(clip), causal=bool(causal), key_pad=bool(pad),  # noqa: E731
                                    key_pad_boolean=bool(pad), mask_min=FMIN)
    return calls, name


def ab(rule, shape, side_a, side_b, iters=120):
    """side = (off_mask, mq_force).  side_a is the library's own choice (0, 0) unless stated."""
    B, H, S, D, dt, causal, clip, int8, pad = shape
    calls, name = build(*shape)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res, names = {0: [], 1: []}, {}
    for rnd in range(10):
        for i, side in enumerate((side_a, side_b)):
            lib.oeh_debug_set_variant(*side)
            names[i] = name()
            for c_ in calls[:3]:
                c_(stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for j in range(iters):
                calls[j % len(calls)](stream)
            e1.record()
            torch.cuda.synchronize()
            if rnd >= 2:
                res[i].append(e0.elapsed_time(e1) * 1e3 / iters)
    lib.oeh_debug_set_variant(0, 0)
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    verdict = "pick ok" if b >= 0.97 * a else "FLIP"
    tag = f"B={B} H={H} S={S} D={D} {dt}{' causal' if causal else ''}{' clip' if clip else ''}{' int8' if int8 else ''}{' pad' if pad else ''}"
    same = " (same kernel: rule not reached)" if names[0] == names[1] else ""
    print(f"{rule:34s} | {tag:44s} | pick {names[0]:30s} {a:7.2f} us | other {names[1]:30s} {b:7.2f} us | other/pick {b / a:5.3f} | {verdict}{same}", flush=True)


def main():
    only = sys.argv[1:]
    want = lambda r: (not only and r != "wide") or any(o in r for o in only)  # noqa: E731  ("wide": an experiment, on request only)
    # 1. one-pass kernel, query blocks per wave: two (128-row workgroups) from 416 workgroups on  (oeh_api.hip: flash_mq)
    if want("flash_mq"):
        for B in (7, 8, 9, 10, 12):  # H=12 S=512: 4 workgroups of 128 rows per head -> 336, 384, 432, 480, 576
            ab("flash_mq >= 416 wgs -> MQ2", (B, 12, 512, 64, "f16", 1, 0, 0, 0), (0, 0), (0, 1 if B * 48 >= 416 else 2))
    # 2. rows of <= 128 keys: the one-pass kernel from 768 of its workgroups on, the full-row kernel below  (flash_eligible)
    if want("short_rows"):
        for B in (32, 48, 64, 96):  # H=12 S=128: B*12 workgroups of 128 rows
            wg = B * 12
            ab("Sk<=128: one-pass >= 768 wgs", (B, 12, 128, 64, "f16", 0, 0, 0, 1), (0, 0), ((2, 0) if wg >= 768 else (256, 0)))
        for B in (64, 128, 192):   # d = 32: the threshold is 1536
            wg = B * 12
            ab("Sk<=128 d=32: one-pass >= 1536", (B, 12, 128, 32, "f16", 0, 0, 0, 0), (0, 0), ((2, 0) if wg >= 1536 else (256, 0)))
    # 3. causal rows that leave the last 128-row workgroup at most half full: full-row kernel  (flash_eligible)
    if want("ragged_causal"):
        for S, B in ((192, 40), (256, 32), (320, 24), (448, 18), (512, 16)):
            half_full = ((S - 1) % 128) < 64
            ab("causal, last wg <= half: full-row", (B, 12, S, 64, "f16", 1, 0, 0, 0), (0, 0), ((256, 0) if half_full else (2, 0)))
    # 4. clipped softmax: the two-pass one-pass form beyond 512 keys, the full-row kernel up to there; d = 128: from 384 keys on
    if want("clip_two_pass"):
        for S, B in ((384, 20), (512, 16)):
            ab("clip: two-pass only > 512 keys", (B, 12, S, 64, "f16", 1, 1, 0, 0), (0, 0), (256, 0))
        for S, B in ((256, 32), (384, 20), (512, 16)):
            ab("clip d=128: two-pass >= 384 keys", (B, 8, S, 128, "f16", 1, 1, 0, 0), (0, 0), ((2, 0) if S >= 384 else (256, 0)))
    # 5. the INT8 chain: two-pass only beyond 512 keys
    if want("int8_two_pass"):
        for S, B in ((384, 20), (512, 16)):
            ab("int8: two-pass only > 512 keys", (B, 12, S, 64, "f16", 1, 0, 1, 0), (0, 0), (256, 0))
    # 6. head dim 128 with clip / INT8: the general kernel instead of the full-row kernel  (d128_general)
    if want("d128_general"):
        for S, B, clip, int8 in ((256, 32, 1, 0), (256, 32, 0, 1), (512, 16, 0, 1)):
            ab("d=128 clip/int8: general kernel", (B, 8, S, 128, "f16", 1, clip, int8, 0), (0, 0), (1 << 11, 0))
    # 7. fp32 storage, rows of <= 128 keys: the full-row fp32 form, except unpadded with >= 768 workgroups  (flash32_eligible)
    if want("fp32_short_rows"):
        for B, pad in ((32, 1), (64, 1), (32, 0), (64, 0), (96, 0)):
            many = (not pad) and B * 12 >= 768
            ab("fp32 Sk<=128: full-row unless many", (B, 12, 128, 64, "f32", 0, 0, 0, pad), (0, 0), ((64, 0) if many else (256, 0)))
    # 8. fp32 storage, causal ragged rows up to 384 keys: the full-row fp32 form
    if want("fp32_ragged_causal"):
        for S, B in ((192, 40), (320, 24), (448, 18)):
            ab("fp32 causal ragged <= 384: full-row", (B, 12, S, 64, "f32", 1, 0, 0, 0), (0, 0), ((256, 0) if S <= 384 else (64, 0)))
    # the 32x32x16 form of the one-pass kernel (oeh_attn_wide.hip) against the production 16x16x32 form
    if want("wide"):
        for B, S, causal in ((16, 512, 1), (16, 512, 0), (8, 1024, 1), (4, 2048, 1), (32, 256, 1), (12, 512, 1)):
            ab("wide (32x32x16) vs one-pass", (B, 12, S, 64, "f16", causal, 0, 0, 0), (0, 0), (4096, 0))
        ab("wide (32x32x16) vs one-pass", (16, 12, 512, 64, "bf16", 1, 0, 0, 0), (0, 0), (4096, 0))
    # 9. the small-shape kernel: fp32 problems of at most 32 rows, >= 256 of them
    if want("small_shape"):
        for B, S in ((224, 28), (384, 32), (224, 48), (48, 28)):
            pick_small = S <= 32 and B * 4 >= 256
            ab("small: fp32, <= 32 rows, >= 256 problems", (B, 4, S, 64, "f32", 0, 0, 0, 0), (0, 0), ((1 << 5, 0) if pick_small else (1 << 10, 0)))


if __name__ == "__main__":
    main()
