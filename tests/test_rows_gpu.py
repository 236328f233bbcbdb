"""GPU parity of the stand-alone row kernels (softmax family, fake-quant, gate, min/max) against the golden
fixtures captured from the reference and against the oracle.  `-m gpu`."""
import json

import numpy as np
import pytest

from oracle import oeh_oracle as O
from tests.conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outeffhop_amd import ops as _ops

    return _ops


def test_softmax_rows_all_registry_keys(ops):
    g = load_golden("softmax_rows.npz")
    x = torch.from_numpy(g["x"]).cuda()
    xm = torch.from_numpy(g["xm"]).cuda()
    for k, (b, ga, et) in O.softmax_table().items():
        spec = ops.SoftmaxSpec(base=b, clip=not (ga == 0.0 and et == 1.0), gamma=ga, eta=et)
        np.testing.assert_allclose(ops.softmax_rows(x, spec).cpu().numpy(), g[f"y[{k}]"], rtol=2e-6, atol=1e-7, err_msg=k)
        if f"ym[{k}]" in g.files:
            np.testing.assert_allclose(ops.softmax_rows(xm, spec).cpu().numpy(), g[f"ym[{k}]"], rtol=2e-6, atol=1e-7, err_msg=k)
            for e in ("known123", "known00", "allmasked", "neg", "big", "below_exp_range", "near_exp_range", "single"):
                got = ops.softmax_rows(torch.from_numpy(g[f"edge_x[{e}]"]).cuda(), spec).cpu().numpy()
                np.testing.assert_allclose(got, g[f"edge_y[{e}][{k}]"], rtol=2e-6, atol=1e-7, err_msg=f"{k}/{e}")


def test_clipped_softmax_callables_positional_as_the_reference_calls_them(ops):
    """models/softmax.py:10-19 as data-first callables: `clipped_softmax(input, dim, eta, gamma)` (cross_models/clip_softmax.py:33), the
    registry's partials, and STanHop's ClipSoftmax / ClipSoftmax_1 modules - all the HIP row kernel, against the reference's rows."""
    from functools import partial

    import outeffhop_amd as oa

    g = load_golden("softmax_rows.npz")
    x = torch.from_numpy(g["x"]).cuda()
    for k, (b, ga, et) in O.softmax_table().items():
        if ga == 0.0 and et == 1.0:
            continue
        f = oa.clipped_softmax1 if b == 1 else oa.clipped_softmax
        np.testing.assert_allclose(f(x, -1, et, ga).cpu().numpy(), g[f"y[{k}]"], rtol=2e-6, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(partial(f, gamma=ga, eta=et)(x, dim=-1).cpu().numpy(), g[f"y[{k}]"], rtol=2e-6, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(oa.SOFTMAX_MAPPING[k](x, dim=-1).cpu().numpy(), g[f"y[{k}]"], rtol=2e-6, atol=1e-7, err_msg=k)
    k = "clippedsoftmax1(-.025:1)"
    np.testing.assert_allclose(oa.ClipSoftmax_1(-1, 1.1, -0.025)(x).cpu().numpy(), g[f"y[{k}]"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(oa.ClipSoftmax(-1, 1.0, -0.025)(x).cpu().numpy(), g["y[clipped(-.025:1)]"], rtol=2e-6, atol=1e-7)
    xt = x.t().contiguous()   # the callables' default dim is 1, as in the reference
    np.testing.assert_allclose(oa.clipped_softmax(xt.t().contiguous().t(), 0, 1.0, -0.025).t().cpu().numpy(), g["y[clipped(-.025:1)]"], rtol=2e-6, atol=1e-7)


def test_softmax_rows_dims_dtypes_and_long_rows(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 40, 17, generator=g)
    want = O.softmax_1(x.numpy(), axis=1)
    np.testing.assert_allclose(ops.softmax_rows(x.cuda(), dim=1).cpu().numpy(), want, rtol=2e-6, atol=1e-7)
    x = torch.randn(2, 20000, generator=g) * 3  # longer than the LDS-staged limit
    np.testing.assert_allclose(ops.softmax_rows(x.cuda()).cpu().numpy(), O.softmax_1(x.numpy()), rtol=3e-6, atol=1e-9)
    xh = torch.randn(64, 512, generator=g).half()
    got = ops.softmax_rows(xh.cuda(), ops.SoftmaxSpec(1, True, -0.025, 1.1)).float().cpu().numpy()
    np.testing.assert_allclose(got, O.apply_softmax(xh.float().numpy(), 1, -0.025, 1.1), atol=1e-3)
    assert ops.softmax_rows(torch.empty(0, 8).cuda()).shape == (0, 8)


def test_fake_quant_bit_exact(ops):
    g = load_golden("fakequant.npz")
    x = torch.from_numpy(g["x"]).cuda()
    for m in json.loads(str(g["meta_json"])):
        t = m["tag"]
        spec = ops.FakeQuantSpec.from_delta(float(g[f"{t}_delta"]), float(g[f"{t}_zero_float"]), m["n_bits"])
        assert np.float32(spec.scale) == g[f"{t}_scale"] and spec.zero_point == float(g[f"{t}_zero_point"])
        xq, idx = ops.fake_quant(x, spec, want_idx=True)
        assert np.array_equal(idx.cpu().numpy().astype(np.float32), g[f"{t}_idx"]), t
        assert np.array_equal(xq.cpu().numpy(), g[f"{t}_xq"]), t
    spec = ops.FakeQuantSpec.from_delta(float(g["pct_delta"]), float(g["pct_zero_float"]))
    xq, idx = ops.fake_quant(torch.from_numpy(g["pct_x"]).cuda(), spec, want_idx=True)
    assert np.array_equal(idx.cpu().numpy().astype(np.float32), g["pct_idx"]) and np.array_equal(xq.cpu().numpy(), g["pct_xq"])
    # a large tensor of softmax1 probabilities against the oracle's true-division quantiser
    p = torch.from_numpy(O.softmax_1(np.random.default_rng(0).standard_normal((4096, 512)).astype(np.float32)))
    d, z = O.quant_range_to_params(*np.percentile(p.numpy(), (0.001, 99.999)))
    want_q, want_i = O.fake_quant(p.numpy(), d, z)
    xq, idx = ops.fake_quant(p.cuda(), ops.FakeQuantSpec.from_delta(d, z), want_idx=True)
    assert np.array_equal(idx.cpu().numpy(), want_i.astype(np.uint8)) and np.array_equal(xq.cpu().numpy(), want_q)


def test_gate_kernels(ops):
    g = load_golden("bert_attn_fp.npz")
    hidden = torch.from_numpy(g["hidden"]).cuda()
    H = 2
    n = 0
    for cj in g["cases_json"]:
        c = json.loads(str(cj))
        sd = {k[len(c["name"]) + 3:]: g[k] for k in g.files if k.startswith(c["name"] + ".w.")}
        kind, gp = O.gate_params_from_state(sd, H)
        if kind not in ("linear", "mlp"):
            continue
        pool = c["gate"].startswith("head_")
        scaling = float(g[f"{c['name']}.gate_scaling_factor"])
        want = O.gate_values(g["hidden"], H, kind, gp, pool) * np.float32(scaling)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        if kind == "linear":
            got = ops.gate_fwd(hidden, H, T(gp["w"]), T(gp["b"]), per_head_pool=pool, scaling=scaling)
        else:
            got = ops.gate_fwd(hidden, H, T(gp["w1"]), T(gp["b1"]), T(gp["w2"]), T(gp["b2"]), per_head_pool=pool, scaling=scaling)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-6, err_msg=c["name"])
        n += 1
    assert n >= 5


def test_minmax(ops):
    x = torch.randn(3, 1000, 77, generator=torch.Generator().manual_seed(9))
    x[1, 5, 5] = -123.5
    x[2, 999, 76] = 77.25
    got = ops.minmax(x.cuda()).cpu().numpy()
    assert got[0] == -123.5 and got[1] == 77.25
    got = ops.minmax(x.half().cuda()).cpu().numpy()
    assert got[0] == -123.5 and got[1] == 77.25


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_percentile_ema_on_device_equals_numpy(ops, dtype):
    """SURVEY 8f-2 / VERDICT r1 #9: the percentile range statistics (range_estimators.py:83-106) by exact radix selection on
    the GPU, numpy's interpolation in float64, running average in device memory - against np.percentile itself, bit for bit,
    on heavy ties, tiny tensors, all-equal data, negative-only data and ranks that fall between distinct values."""
    g = torch.Generator().manual_seed(5)
    cases = [torch.randn(1000003, generator=g) * 3.0, torch.randn(70001, generator=g).round(),  # heavy ties
             torch.full((4097,), -2.5), -torch.rand(50000, generator=g) - 1.0, torch.randn(7, generator=g), torch.randn(1, generator=g),
             torch.cat([torch.zeros(100000), torch.randn(33, generator=g) * 100.0]), torch.randn(2, 12, 256, 256, generator=g)]
    for n, x in enumerate(cases):
        x = x.to(dtype)
        ref = x.float().numpy().reshape(-1)
        for (qlo, qhi) in ((0.001, 99.999), (1.0, 99.0), (0.0, 100.0), (50.0, 50.0)):
            want = np.percentile(ref, (qlo, qhi))
            st = torch.zeros(2, dtype=torch.float64, device="cuda")
            ops.percentile_ema(x.cuda(), qlo, qhi, st, first=True)
            got = st.cpu().numpy()
            assert got.dtype == np.float64 and np.array_equal(got, want), (n, dtype, qlo, qhi, got, want)
    # the running average over four batches: the reference's trajectory (golden, float64)
    gold = load_golden("range_estimators.npz")
    st = torch.zeros(2, dtype=torch.float64, device="cuda")
    for i in range(4):
        ops.percentile_ema(torch.from_numpy(gold[f"batch{i}"]).cuda(), 100 - 99.999, 99.999, st, momentum=0.9, first=(i == 0))
        np.testing.assert_allclose(st.cpu().numpy(), gold["running_pct_traj"][i][:2], rtol=1e-15, atol=0)


def test_fake_quant_with_a_device_resident_range(ops):
    """oeh_fake_quant_range derives the grid in the kernel from a float64 (x_min, x_max) pair exactly as set_quant_range does:
    same bits as the host-descriptor path (FakeQuantSpec.from_delta of the same range)."""
    g = torch.Generator().manual_seed(6)
    for (lo, hi) in ((-3.2187, 4.000123), (0.0, 0.99871), (-7.5, -1.0), (0.25, 9.0), (-1e-12, 1e-12)):
        for dtype in (torch.float32, torch.float16):
            x = (torch.randn(100000, generator=g) * 3.0).to(dtype).cuda()
            rng = torch.tensor([lo, hi], dtype=torch.float64, device="cuda")
            got = ops.fake_quant_range(x, rng, 8, 1e-8)
            x_min, x_max = min(lo, 0.0), max(hi, 1e-8)
            delta = (x_max - x_min) / 255.0
            want = ops.fake_quant(x, ops.FakeQuantSpec.from_delta(delta, -x_min / delta, 8, 1e-8))
            assert torch.equal(got, want), (lo, hi, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_quantize_heads_i8_equals_fake_quant_indices(ops, dtype):
    """oeh_quantize_heads_i8 (the producer side of the INT8-storage attention core): the centred indices idx - 128 and the
    dequantised values equal oeh_fake_quant's (bit-exact quantiser, pinned to the reference by tests/golden/fakequant.npz), in
    the (B,S,H*64) layout and transposed per head to (B,H,64,S); strided input rows (a slice of a fused q/k/v GEMM output)."""
    g = torch.Generator().manual_seed(9)
    B, S, H = 3, 112, 5
    big = (torch.randn(B, S, 3 * H * 64, generator=g) * 2.0).to(dtype).cuda()
    for n, (scale, zp) in enumerate(((0.031, 131.0), (0.02, 0.0), (0.05, 255.0))):
        x = big[..., n * H * 64:(n + 1) * H * 64]  # strided rows
        spec = ops.FakeQuantSpec(scale, zp)
        want_y, want_idx = ops.fake_quant(x, spec, want_idx=True)
        want_c = (want_idx.to(torch.int16) - 128).to(torch.int8)
        got, y = ops.quantize_heads_i8(x, spec, H, transpose=False, want_values=True)
        assert got.shape == (B, H, S, 64) and torch.equal(got, want_c.view(B, S, H, 64).permute(0, 2, 1, 3)) and torch.equal(y, want_y)
        got_t = ops.quantize_heads_i8(x, spec, H, transpose=True)
        assert got_t.shape == (B, H, 64, S) and got_t.is_contiguous()
        assert torch.equal(got_t, want_c.view(B, S, H, 64).permute(0, 2, 3, 1))
