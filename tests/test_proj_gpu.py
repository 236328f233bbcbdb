"""`oeh_proj_quant_i8` - the q / k / v projections of a QuantLinear model as ONE GEMM with the output quantisers in its epilogue
(quantized_opt.py:67-75, quantized_bert.py:236-238, hijacker.py:78-127) - against the exact arithmetic: float64
products of the same operands, the reference's quantiser formula (uniform_quantizers.py:114-148: clamp(round(x / scale) +
zero_point, 0, 255)), and against the path it replaces (library pair GEMM + `oeh_quantize_heads_i8`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _exact_indices(v64, spec, B, S, H, transpose):
    idx = (torch.clamp(torch.round(v64 / float(spec.scale)) + spec.zero_point, 0, 255) - 128).to(torch.int32).view(B, S, H, 64).permute(0, 2, 1, 3)
    return (idx.permute(0, 1, 3, 2) if transpose else idx).contiguous()


def _grid(v64):
    lo, hi = v64.min().item() * 0.9, v64.max().item() * 0.9  # (a little clipping at both ends)
    sc = np.float32((hi - lo) / 255.0)
    return float(sc), float(np.clip(np.rint(-lo / sc), 0, 255))


@pytest.mark.parametrize("B,S,H,K,want", [(16, 512, 12, 768, True), (4, 128, 12, 768, False), (3, 48, 2, 64, True), (5, 80, 12, 768, True), (2, 16, 1, 32, True),
                                          (7, 144, 5, 320, False), (32, 128, 12, 768, True), (25, 160, 12, 64, False), (17, 480, 12, 96, True)])   # (one 128 x 288 tile per CU - the LOOP == 1 kernel; ragged rows, two K steps; the last: the pipelined two-per-CU loop with a ragged last row tile, three K steps)
def test_pair_gemm_with_quantiser_epilogue_vs_exact_arithmetic(B, S, H, K, want):
    """fp32 activations as operand pairs: every index within one step of the exact one and all but a few in 10^5 equal to it
    (the value sits on a rounding boundary to within the fp32 accumulation error: the library GEMM + quantiser pass it replaces
    shows the same rate), the written values are scale * (index - zero_point) exactly, all layouts, ragged tiles."""
    from outeffhop_amd import ops

    torch.manual_seed(B * 1000 + S)
    E, M = H * 64, B * S
    x = torch.randn(B, S, K, device="cuda")
    x[..., ::37] *= 30.0
    wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
    bias = torch.randn(3 * E, device="cuda") * 0.1
    alphas = [0.003, 0.0025, 0.002]
    pairs = ops.split_pairs(x.view(M, K))
    ref64 = x.view(M, K).double() @ wi.double().t()
    vals = [ref64[:, n * E:(n + 1) * E] * alphas[n] + bias[n * E:(n + 1) * E].double() for n in range(3)]
    specs = [ops.FakeQuantSpec(*_grid(v)) for v in vals]
    new = ops.proj_quant_i8(pairs, wi, bias, B, S, [(alphas[n], specs[n], n == 2, n > 0 and want) for n in range(3)], pairs=True)
    new32 = ops.proj_quant_i8(x.view(M, K), wi, bias, B, S, [(alphas[n], specs[n], n == 2, n > 0 and want) for n in range(3)], pairs=True)
    ww3 = torch.cat([wi, wi * 2.0 ** -11], dim=1).t().contiguous()
    acc3 = torch.mm(pairs, ww3, out_dtype=torch.float32).view(B, S, 3 * E)
    for n in range(3):
        idx, y = new[n] if (n > 0 and want) else (new[n], None)
        assert idx.shape == ((B, H, 64, S) if n == 2 else (B, H, S, 64)) and idx.dtype == torch.int8
        got = idx.contiguous().to(torch.int32)
        ex = _exact_indices(vals[n], specs[n], B, S, H, n == 2)
        d = (got - ex).abs()
        old = ops.quantize_heads_i8(acc3[..., n * E:(n + 1) * E], specs[n], H, transpose=(n == 2), alpha=alphas[n], bias=bias[n * E:(n + 1) * E].contiguous())
        d_old = (old.contiguous().to(torch.int32) - ex).abs()
        rate, rate_old = float((d != 0).float().mean()), float((d_old != 0).float().mean())
        print(f"B={B} S={S} H={H} K={K} [{'qkv'[n]}]: index != exact {rate:.1e} (library GEMM + quantiser pass: {rate_old:.1e}), max {int(d.max())} step")
        assert int(d.max()) <= 1 and rate <= max(3e-5, 3.0 * rate_old)
        # ... and from the fp32 activations themselves (the modules' path; round 5: the residual operand unscaled, oeh_common.h: split8_raw_scaled): the same bar
        got32 = (new32[n][0] if (n > 0 and want) else new32[n]).contiguous().to(torch.int32)
        d32 = (got32 - ex).abs()
        rate32 = float((d32 != 0).float().mean())
        print(f"   from the fp32 activations: index != exact {rate32:.1e}, max {int(d32.max())} step")
        assert int(d32.max()) <= 1 and rate32 <= max(3e-5, 3.0 * rate_old)
        if y is not None:
            rows = got.permute(0, 1, 3, 2) if n == 2 else got          # (B, H, S, 64)
            want_y = np.float32(specs[n].scale) * (rows.permute(0, 2, 1, 3).reshape(B, S, E).float() + 128.0 - specs[n].zero_point)
            assert y.shape == (B, S, E) and y.dtype == torch.float32 and torch.equal(y, want_y)


@pytest.mark.parametrize("pairs", [False, True])
def test_projection_gemm_is_exact_on_exactly_representable_sums(pairs):
    """Small integers on both sides: every partial sum is an integer below 2^24, so any accumulation order gives the same fp32
    accumulator and the indices must equal the exact ones bit for bit (fp16 activations, and operand pairs of the same values)."""
    from outeffhop_amd import ops

    torch.manual_seed(5)
    B, S, H, K = 3, 160, 3, 96
    E, M = H * 64, B * S
    xi = torch.randint(-40, 41, (M, K), device="cuda")
    wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
    bias = torch.randint(-50, 51, (3 * E,), device="cuda").float()
    a = ops.split_pairs(xi.float()) if pairs else xi.to(torch.float16)
    ref64 = xi.double() @ wi.double().t()
    alphas = [0.5, 0.25, 1.0]   # powers of two: alpha * acc + bias is exact too
    vals = [ref64[:, n * E:(n + 1) * E] * alphas[n] + bias[n * E:(n + 1) * E].double() for n in range(3)]
    specs = [ops.FakeQuantSpec(float(np.float32(2.0 ** (7 + n))), float(100 + 20 * n)) for n in range(3)]  # (power-of-two steps: exact quotients)
    out = ops.proj_quant_i8(a, wi, bias, B, S, [(alphas[n], specs[n], n == 2, n == 1) for n in range(3)], pairs=pairs)
    for n in range(3):
        idx = out[n][0] if n == 1 else out[n]
        assert torch.equal(idx.contiguous().to(torch.int32), _exact_indices(vals[n], specs[n], B, S, H, n == 2))


def test_projection_gemm_refuses_what_it_does_not_do():
    from outeffhop_amd import _lib, ops

    a = torch.zeros(32, 2 * 40, dtype=torch.float16, device="cuda")
    w = torch.zeros(3 * 64, 40, dtype=torch.float16, device="cuda")
    b = torch.zeros(3 * 64, device="cuda")
    sp = ops.FakeQuantSpec(0.1, 128.0)
    with pytest.raises(_lib.OehError) as e:
        ops.proj_quant_i8(a, w, b, 2, 16, [(1.0, sp, n == 2, False) for n in range(3)], pairs=True)  # K % 32 != 0
    assert e.value.code == -95


def test_projection_gemm_random_shapes_against_the_library_path():
    """Random (B, S, H, K), one to three segments, both operand forms, values on / off per segment, both tile shapes (the library
    picks by problem size; the diagnostic switch forces the other): every index within one step of the library GEMM + quantiser
    pass on the same operands, all but a few in 10^5 equal; values consistent with the indices."""
    import os
    from outeffhop_amd import ops

    rng = np.random.default_rng(7)
    for trial in range(24):
        B, S = int(rng.integers(1, 9)), 16 * int(rng.integers(1, 20))
        H, K = int(rng.integers(1, 7)), 32 * int(rng.integers(1, 13))
        nseg = int(rng.integers(1, 4))
        pairs = bool(rng.integers(0, 2))
        E, M = H * 64, B * S
        torch.manual_seed(trial)
        x = torch.randn(M, K, device="cuda")
        wi = torch.randint(-128, 128, (nseg * E, K), device="cuda").to(torch.float16)
        bias = torch.randn(nseg * E, device="cuda") * 0.2
        a = ops.split_pairs(x) if pairs else x.to(torch.float16)
        ww = (torch.cat([wi, wi * 2.0 ** -11], dim=1) if pairs else wi).t().contiguous()
        acc = torch.mm(a, ww, out_dtype=torch.float32).view(B, S, nseg * E)
        alphas = [float(rng.uniform(1e-3, 4e-3)) for _ in range(nseg)]
        segs, specs = [], []
        for n in range(nseg):
            v = acc[..., n * E:(n + 1) * E] * alphas[n] + bias[n * E:(n + 1) * E]
            specs.append(ops.FakeQuantSpec(*_grid(v.double())))
            segs.append((alphas[n], specs[n], bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
        new = ops.proj_quant_i8(a, wi, bias, B, S, segs, pairs=pairs)
        for n, (al, sp, tr, want) in enumerate(segs):
            old = ops.quantize_heads_i8(acc[..., n * E:(n + 1) * E], sp, H, transpose=tr, want_values=want, alpha=al, bias=bias[n * E:(n + 1) * E].contiguous())
            io, yo = old if want else (old, None)
            inw, yn = new[n] if want else (new[n], None)
            d = (io.contiguous().to(torch.int32) - inw.contiguous().to(torch.int32)).abs()
            assert int(d.max()) <= 1 and float((d != 0).float().mean()) <= 1e-4, (trial, B, S, H, K, nseg, pairs, n, int(d.max()), float((d != 0).float().mean()))
            if want:
                rows = inw.permute(0, 1, 3, 2) if tr else inw
                want_y = np.float32(sp.scale) * (rows.permute(0, 2, 1, 3).reshape(B, S, E).float() + 128.0 - sp.zero_point)
                assert torch.equal(yn, want_y)


@pytest.mark.parametrize("mag", [1.0e-4, 1.0e-3, 0.05, 1.0, 450.0])
def test_fp32_activations_of_any_magnitude_vs_exact_arithmetic(mag):
    """The in-kernel split of fp32 activations (pairs == 2) keeps the residual unscaled, on 32 x (oeh_common.h: split8_raw_scaled): 22 bits of every
    value down to |x| = 2^-8, an absolute 2^-30 below, range |x| <= 2 047 (11 bits up to 4 094).  A whole tensor of tiny values (the case an unscaled residual WITHOUT the
    pre-scale would lose: fp16 subnormals) and one that fills the range: the indices against float64 arithmetic, the same bar as at magnitude 1."""
    from outeffhop_amd import ops

    torch.manual_seed(int(mag * 1e4) + 3)
    B, S, H, K = (4, 256, 12, 768) if mag != 0.05 else (32, 128, 12, 768)   # (one case on the one-workgroup-per-CU kernel)
    E, M = H * 64, B * S
    x = (torch.randn(M, K, device="cuda") * mag).clamp_(-2040.0, 2040.0)
    wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
    bias = torch.randn(3 * E, device="cuda") * 0.1
    alphas = [0.003 / mag, 0.0025 / mag, 0.002 / mag]
    ref64 = x.double() @ wi.double().t()
    vals = [ref64[:, n * E:(n + 1) * E] * alphas[n] + bias[n * E:(n + 1) * E].double() for n in range(3)]
    specs = [ops.FakeQuantSpec(*_grid(v)) for v in vals]
    got = ops.proj_quant_i8(x, wi, bias, B, S, [(alphas[n], specs[n], n == 2, False) for n in range(3)], pairs=True)
    for n in range(3):
        ex = _exact_indices(vals[n], specs[n], B, S, H, n == 2)
        d = (got[n].contiguous().to(torch.int32) - ex).abs()
        rate = float((d != 0).float().mean())
        print(f"|x| ~ {mag:g} [{'qkv'[n]}]: index != exact {rate:.1e}, max {int(d.max())} step")
        assert int(d.max()) <= 1 and rate <= (3e-5 if mag >= 1e-3 else 3e-4)   # (|x| ~ 1e-4 is 5 bits below 2^-8: 17 bits)


@pytest.mark.parametrize("B,S,H,K", [(16, 512, 12, 768), (4, 128, 12, 768), (3, 48, 2, 64), (5, 80, 12, 768), (32, 128, 12, 768), (25, 160, 12, 96)])
def test_fp32_activations_split_inside_the_kernel_equal_the_operand_pairs(B, S, H, K):
    """`a` as the fp32 activation matrix (split into fp16 operands when a wave reads its fragments) against the same call on `oeh_split_pairs`'
    output.  Round 5: the in-kernel split keeps the residual UNSCALED, on 32 x (oeh_common.h: split8_raw_scaled - the matrix core takes fp16 subnormals
    exactly, tools/probe/mix_probe.hip), `oeh_split_pairs` keeps its documented [hi | lo 2^11] format: the two represent x to 2^-30 absolute / 2^-22
    relative respectively, so an index may differ where a value sits on a rounding boundary to within that - never by more than one step, in at most 2e-5
    of the outputs (measured ~1e-6; rounds 1-4: bit-identical, with both paths on the scaled pair).  Values beyond the range: the pair format saturates
    at 65504 (+ 32), the in-kernel one at 4 094 - both finite; those rows are compared for finiteness only."""
    from outeffhop_amd import ops

    torch.manual_seed(11 + S)
    E, M = H * 64, B * S
    x = torch.randn(M, K, device="cuda")
    x[:, ::53] *= 40.0
    x[0, 1], x[1, 2] = 7.0e4, -3.0e5     # beyond fp16
    wi = torch.randint(-128, 128, (3 * E, K), device="cuda").to(torch.float16)
    bias = torch.randn(3 * E, device="cuda") * 0.1
    specs = [ops.FakeQuantSpec(0.03, 131.0), ops.FakeQuantSpec(0.035, 124.0), ops.FakeQuantSpec(0.03, 128.0)]
    segs = [(0.003, specs[n], n == 2, n > 0) for n in range(3)]
    flat = lambda r: [t_ for o in r for t_ in (o if isinstance(o, tuple) else (o,))]  # noqa: E731
    a = flat(ops.proj_quant_i8(ops.split_pairs(x), wi, bias, B, S, segs, pairs=True))
    b = flat(ops.proj_quant_i8(x, wi, bias, B, S, segs, pairs=True))
    assert len(a) == len(b) == 5
    steps = [1.0, 1.0, 0.035, 1.0, 0.03]   # q idx | k idx, k values | v idx (transposed), v values
    for n, (p_, q_) in enumerate(zip(a, b)):
        assert p_.shape == q_.shape and p_.dtype == q_.dtype
        if p_.dtype != torch.int8:
            assert torch.isfinite(q_).all()
        if n == 3:   # v indices, (B, H, 64, S): token rows 0 and 1 of batch 0 are columns 0 and 1 of every head's tile
            pm, qm = p_[0, :, :, 2:].float(), q_[0, :, :, 2:].float()
            rest = (p_[1:].float(), q_[1:].float()) if B > 1 else None
        else:        # (B, S, E) row-major views: drop rows 0 and 1 of batch 0
            pv = p_.permute(0, 2, 1, 3).reshape(B, S, E) if p_.dim() == 4 else p_.reshape(B, S, E)
            qv = q_.permute(0, 2, 1, 3).reshape(B, S, E) if q_.dim() == 4 else q_.reshape(B, S, E)
            pm, qm = pv[0, 2:].float(), qv[0, 2:].float()
            rest = (pv[1:].float(), qv[1:].float()) if B > 1 else None
        d = (pm - qm).abs()
        cnt, tot, worst = int((d > 0).sum()), d.numel(), float(d.max()) if d.numel() else 0.0
        if rest is not None:
            d2 = (rest[0] - rest[1]).abs()
            cnt, tot, worst = cnt + int((d2 > 0).sum()), tot + d2.numel(), max(worst, float(d2.max()))
        assert worst <= steps[n] * 1.0001 and cnt <= max(2, 2e-5 * tot), f"output {n}: {cnt} of {tot} differ, max {worst / steps[n]:.2f} steps"


@pytest.mark.parametrize("M,N,K", [(8192, 768, 768), (400, 128, 64), (4096, 768, 768), (48, 64, 128)])
def test_int8_activations_on_the_integer_matrix_cores_equal_the_fp16_integer_form(M, N, K):
    """out_proj on the context quantiser's indices: int8 centred indices c = idx - 128 against the int8 weights (v_mfma_i32_16x16x64_i8,
    exact int32 sums) with acc_add = (128 - zp) * column sums, against the same projection on the integers idx - zp carried as fp16
    (exact products, fp32 sums exact below 2^24): the same fake-quantised values bit for bit."""
    from outeffhop_amd import ops

    torch.manual_seed(M + N)
    zp = 117
    idx = torch.randint(90, 150, (M, K), device="cuda")           # (sums of (idx - zp) * w stay below 2^24 for K = 768)
    wi = torch.randint(-128, 128, (N, K), device="cuda")
    bias = torch.randn(N, device="cuda") * 0.1
    spec = ops.FakeQuantSpec(0.02, 131.0)
    alpha = 3.1e-5
    y16 = ops.proj_quant_values((idx - zp).to(torch.float16), wi.to(torch.float16), bias, alpha, spec, pairs=False)
    add = ((128 - zp) * wi.sum(dim=1)).to(torch.int32).contiguous()
    y8 = ops.proj_quant_values((idx - 128).to(torch.int8), wi.to(torch.int8), bias, alpha, spec, pairs=False, acc_add=add)
    exact = (idx - zp).double() @ wi.double().t()
    assert float(exact.abs().max()) < 2 ** 24
    assert torch.equal(y8, y16)
