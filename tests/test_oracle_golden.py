"""Pin the CPU oracle (oracle/oeh_oracle.py) against fixtures captured from the reference itself
(tests/golden/make_golden.py).  CPU only; no reference import, no HIP."""
import json
import math

import numpy as np
import pytest

from oracle import oeh_oracle as O
from tests.conftest import load_golden

ELEM = dict(rtol=2e-6, atol=1e-7)   # elementwise chains: exp() implementations differ by <= 1-2 ulp
MM = dict(rtol=1e-5, atol=2e-6)     # chains containing fp32 matmuls (BLAS accumulation order)


def test_softmax_table_matches_reference_registry():
    g = load_golden("softmax_rows.npz")
    tbl = O.softmax_table()
    ref_keys = [k for k in g["all_keys_in_order"] if k != "entmax"]
    assert list(tbl.keys()) == ref_keys and len(tbl) == 39
    for k, b, ga, et in zip(g["keys"], g["key_base"], g["key_gamma"], g["key_eta"]):
        assert tbl[str(k)] == (int(b), float(ga), float(et)), k
    assert tbl["clippedsoftmax1(-.025:1)"] == (1, -0.025, 1.1)      # eta quirk, softmax.py:61
    assert tbl["clipped(-.005:1.005)"] == (0, -0.003, 1.005)        # gamma quirk, softmax.py:57


def test_softmax_rows_all_keys():
    g = load_golden("softmax_rows.npz")
    tbl = O.softmax_table()
    for k, (b, ga, et) in tbl.items():
        np.testing.assert_allclose(O.apply_softmax(g["x"], b, ga, et), g[f"y[{k}]"], err_msg=k, **ELEM)


def test_softmax_masked_and_edge_rows():
    g = load_golden("softmax_rows.npz")
    tbl = O.softmax_table()
    edges = ["known123", "known00", "allmasked", "neg", "big", "below_exp_range", "near_exp_range", "single"]
    for k in ("vanilla", "softmax1", "clipped(-.025:1)", "clippedsoftmax1(-.025:1)", "clipped(0:1.03)"):
        b, ga, et = tbl[k]
        np.testing.assert_allclose(O.apply_softmax(g["xm"], b, ga, et), g[f"ym[{k}]"], err_msg=k, **ELEM)
        for e in edges:
            got = O.apply_softmax(g[f"edge_x[{e}]"], b, ga, et)
            np.testing.assert_allclose(got, g[f"edge_y[{e}][{k}]"], err_msg=f"{k}/{e}", **ELEM)
    # known answers quoted in SURVEY 8c
    np.testing.assert_allclose(O.softmax_1(np.array([0.0, 0.0])), [1 / 3, 1 / 3], rtol=1e-6)
    np.testing.assert_allclose(O.softmax_1(np.array([1.0, 2.0, 3.0])), [0.0871443227, 0.2368828356, 0.6439142823], rtol=1e-6)
    # fully masked row -> exactly zero under softmax1 (vanilla gives uniform)
    assert np.all(O.softmax_1(g["edge_x[allmasked]"]) == 0.0)
    assert np.all(g["edge_y[allmasked][softmax1]"] == 0.0)
    assert np.all(O.softmax_1(g["edge_x[below_exp_range]"]) == 0.0)


def test_fp16_eager_deviation_is_documented():
    """Reference in pure-fp16 eager math zeroes rows with max < -11.09 (exp(-m) overflows in fp16);
    the oracle / kernel contract is fp32 math on the fp16-rounded inputs."""
    g = load_golden("softmax_rows.npz")
    ref16 = g["fp16_y_softmax1"].astype(np.float32)
    mine = O.softmax_1(g["fp16_x"].astype(np.float32))
    assert np.all(ref16[0] == 0.0) and np.all(mine[0] > 0) and np.all(mine[0] < 1e-4)
    np.testing.assert_allclose(mine[1:], ref16[1:], atol=1e-3)


def test_fake_quant_bit_exact():
    g = load_golden("fakequant.npz")
    for m in json.loads(str(g["meta_json"])):
        t = m["tag"]
        scale, zp, qmax = O.fq_grid(g[f"{t}_delta"], g[f"{t}_zero_float"], m["n_bits"])
        assert scale == g[f"{t}_scale"] and zp == g[f"{t}_zero_point"], t
        idx = O.fq_index(g["x"], scale, zp, qmax)
        assert np.array_equal(idx, g[f"{t}_idx"]), t
        assert np.array_equal(O.fq_dequant(idx, scale, zp), g[f"{t}_xq"]), t
        d, z = O.quant_range_to_params(m["lo"], m["hi"], m["n_bits"])
        assert np.float32(d) == np.float32(g[f"{t}_delta"]) and np.float32(z) == np.float32(g[f"{t}_zero_float"]), t
    # float64 calibration scalars (np.percentile) -> same grid after fp32 rounding
    d, z = O.quant_range_to_params(np.float64(g["pct_lo"]), np.float64(g["pct_hi"]))
    assert d == g["pct_delta"] and z == g["pct_zero_float"] and d.dtype == np.float64
    xq, idx = O.fake_quant(g["pct_x"], g["pct_delta"], g["pct_zero_float"])
    assert np.array_equal(idx, g["pct_idx"]) and np.array_equal(xq, g["pct_xq"])
    assert np.array_equal(O.sym_weight_quant(g["sym_w"]), g["sym_wq"])


def test_running_minmax_estimator():
    g = load_golden("range_estimators.npz")
    for tag, kw in (("pct", dict(percentile=99.999)), ("minmax", dict()), ("pct99", dict(percentile=99.0))):
        est = O.RunningMinMax(**kw)
        for i in range(4):
            lo, hi = est.update(g[f"batch{i}"])
            d, z = O.quant_range_to_params(lo, hi)
            np.testing.assert_allclose([lo, hi, d, z], g[f"running_{tag}_traj"][i], rtol=1e-6, atol=1e-9, err_msg=f"{tag}/{i}")


def _state(g, prefix, case=None):
    sd = {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}
    if case is not None:
        sd.update({k[len(case) + 3:]: g[k] for k in g.files if k.startswith(case + ".w.")})
    return sd


def _case_kwargs(c, tbl):
    b, ga, et = tbl[c["softmax"]]
    return dict(base=b, gamma=ga, eta=et, clip=not (ga == 0.0 and et == 1.0), per_head_pool=c["gate"].startswith("head_"))


def test_bert_module_fp():
    g = load_golden("bert_attn_fp.npz")
    tbl = O.softmax_table()
    for cj in g["cases_json"]:
        c = json.loads(str(cj))
        sd = _state(g, "w.", c["name"])
        kw = _case_kwargs(c, tbl)
        gs = float(g[f"{c['name']}.gate_scaling_factor"])
        ctx, ex = O.bert_self_attention(sd, g["hidden"], 2, mask=g["mask"], gate_scaling=gs, want=("probs",), **kw)
        np.testing.assert_allclose(ex["probs"], g[f"{c['name']}.probs"], err_msg=c["name"], **MM)
        np.testing.assert_allclose(ctx, g[f"{c['name']}.ctx"], err_msg=c["name"], **MM)
        ctx = O.bert_self_attention(sd, g["hidden"], 2, gate_scaling=gs, **kw)
        np.testing.assert_allclose(ctx, g[f"{c['name']}.ctx_nomask"], err_msg=c["name"], **MM)
    # ctor alpha=4, max_seq_length=32 -> clipped softmax gamma=-4/32, eta=1 (bert_attention.py:89-92)
    ctx = O.bert_self_attention(_state(g, "w."), g["hidden"], 2, mask=g["mask"], base=0, gamma=-4.0 / 32, eta=1.0, clip=True)
    np.testing.assert_allclose(ctx, g["alpha4.ctx"], **MM)
    assert not g["skip.ctx"].any()
    # the reference's pure-fp16 module agrees with fp32-math-on-fp16-inputs to ~1e-3 (documented contract)
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in _state(g, "w.").items()}
    ctx = O.bert_self_attention(sd16, g["hidden"].astype(np.float16).astype(np.float32), 2, mask=g["mask"])
    np.testing.assert_allclose(ctx, g["half.ctx"].astype(np.float32), atol=4e-3)


def test_opt_module_fp():
    g = load_golden("opt_attn_fp.npz")
    tbl = O.softmax_table()
    for cj in g["cases_json"]:
        c = json.loads(str(cj))
        sd = _state(g, "w.", c["name"])
        kw = _case_kwargs(c, tbl)
        gs = float(g[f"{c['name']}.gate_scaling_factor"])
        out, ex = O.opt_attention(sd, g["hidden"], 2, mask=g["mask"], gate_scaling=gs, want=("probs",), **kw)
        np.testing.assert_allclose(ex["probs"], g[f"{c['name']}.probs"], err_msg=c["name"], **MM)
        np.testing.assert_allclose(out, g[f"{c['name']}.out"], err_msg=c["name"], **MM)
        out = O.opt_attention(sd, g["hidden"], 2, gate_scaling=gs, **kw)
        np.testing.assert_allclose(out, g[f"{c['name']}.out_nomask"], err_msg=c["name"], **MM)
    # alpha=12, max_seq_length=32, attn_softmax="softmax1": the `is "softmax1"` test (opt_attention.py:73)
    name = str(g["alpha12.softmax_fn_name"])
    base = 1 if name == "clipped_softmax1" else 0
    out = O.opt_attention(_state(g, "w."), g["hidden"], 2, mask=g["mask"], base=base, gamma=-12.0 / 32, eta=1.0, clip=True)
    np.testing.assert_allclose(out, g["alpha12.out"], **MM)
    assert str(g["half_softmax1_error"]) == "TypeError"  # fp16 OPT + softmax1 cannot run in the reference


def test_core_cases_fp16_inputs():
    g = load_golden("core_attn.npz")
    tbl = O.softmax_table()
    q, k, v = g["q"], g["k"], g["v"]
    d = q.shape[-1]
    for sm in ("softmax1", "vanilla", "clippedsoftmax1(-.025:1)"):
        b, ga, et = tbl[sm]
        clip = not (ga == 0.0 and et == 1.0)
        ctx, ex = O.attn_core(q, k, v, scale=math.sqrt(d), scale_is_divisor=True, base=b, gamma=ga, eta=et, clip=clip,
                              pad_mask=g["pad_mask"].reshape(q.shape[0], -1), want=("probs",))
        np.testing.assert_allclose(ex["probs"], g[f"bert[{sm}].probs"], **MM)
        np.testing.assert_allclose(ctx, g[f"bert[{sm}].ctx"], **MM)
        qs = (q * np.float32(d ** -0.5)).astype(np.float16).astype(np.float32)
        ctx, ex = O.attn_core(qs, k, v, base=b, gamma=ga, eta=et, clip=clip, full_mask=g["opt_mask"], clamp_min=True, want=("probs",))
        np.testing.assert_allclose(ex["probs"], g[f"opt[{sm}].probs"], **MM)
        np.testing.assert_allclose(ctx, g[f"opt[{sm}].ctx"], **MM)
        # causal flag + key-padding vector == the materialised (B,1,T,S) HF mask
        padvec = g["opt_mask"][:, 0, -1, :]
        ctx2 = O.attn_core(qs, k, v, base=b, gamma=ga, eta=et, clip=clip, pad_mask=padvec, causal=True, clamp_min=True)
        np.testing.assert_allclose(ctx2, g[f"opt[{sm}].ctx"], **MM)


def _fq_from_golden(g, prefix):
    def one(n):
        return (np.float64(g[f"{prefix}.q.{n}.activation_quantizer.delta"]),
                np.float64(g[f"{prefix}.q.{n}.activation_quantizer.zero_float"]))
    return dict(scores=one("attn_scores_act_quantizer"), probs=one("attn_probs_act_quantizer"), ctx=one("context_act_quantizer"))


@pytest.mark.parametrize("fam", ["bert", "opt"])
def test_int8_core_from_captured_qkv(fam):
    """Given the reference's own (already fake-quantised) Q/K/V, the oracle core reproduces the three
    quantiser index tensors and the module output.  Index flips can only come from fp32 matmul
    accumulation order, so they are rare and +-1; everything else must match."""
    g = load_golden("int8_attn.npz")
    tbl = O.softmax_table()
    for m in json.loads(str(g["meta_json"])):
        pre = f"{fam}{m['tag']}"
        b, ga, et = tbl[m["softmax"]]
        clip = not (ga == 0.0 and et == 1.0)
        fq = _fq_from_golden(g, pre)
        sd = _state(g, pre + ".w.")
        H = 2
        ql, kl, vl = g[f"{pre}.q_lin"], g[f"{pre}.k_lin"], g[f"{pre}.v_lin"]
        # quantiser boundary: same pre-quant input => identical indices (bit-exact)
        for name in ("scores", "probs", "ctx"):
            _, idx = O.fake_quant(g[f"{pre}.{name}.in"], *fq[name])
            assert np.array_equal(idx.astype(np.uint8), g[f"{pre}.{name}.idx"]), (pre, name)
        kind, gp = O.gate_params_from_state(sd, H)
        gate = None if kind is None else O.gate_values(g["eval"], H, kind, gp, False)
        q, k, v = O.split_heads(ql, H), O.split_heads(kl, H), O.split_heads(vl, H)
        if fam == "bert":
            ctx, ex = O.attn_core(q, k, v, scale=8.0, scale_is_divisor=True, base=b, gamma=ga, eta=et, clip=clip,
                                  pad_mask=g["bert_mask"].reshape(2, -1), gate=gate, fq_scores=fq["scores"], fq_probs=fq["probs"],
                                  want=("scores_idx", "probs_idx"))
            out = O.merge_heads(ctx)
            out, cidx = O.fake_quant(out, *fq["ctx"])
        else:
            q = (q * np.float32(64 ** -0.5)).astype(np.float32)
            ctx, ex = O.attn_core(q, k, v, base=b, gamma=ga, eta=et, clip=clip, full_mask=g["opt_mask"], clamp_min=True, gate=gate,
                                  fq_scores=fq["scores"], fq_probs=fq["probs"], fq_ctx=fq["ctx"], ctx_quant_before_gate=True,
                                  want=("scores_idx", "probs_idx", "ctx_idx"))
            cidx = ex["ctx_idx"]
        for name, got in (("scores", ex["scores_idx"]), ("probs", ex["probs_idx"]), ("ctx", cidx)):
            ref = g[f"{pre}.{name}.idx"].reshape(got.shape)
            diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
            assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (pre, name, diff.max(), (diff != 0).mean())
        if fam == "bert":
            ref = g[f"{pre}.out"]
            step = np.float32(fq["ctx"][0])
            bad = np.abs(out - ref) > 1e-5
            assert bad.mean() < 2e-3 and np.abs(out - ref).max() <= 1.01 * step


def test_eager_torch_matches_golden():
    """oracle/eager_torch.py (the CPU-baseline op chain) against the same reference fixtures and the numpy oracle."""
    import torch

    from oracle import eager_torch as E

    g = load_golden("core_attn.npz")
    tbl = O.softmax_table()
    q, k, v = (torch.from_numpy(g[n]) for n in ("q", "k", "v"))
    for sm in ("softmax1", "vanilla", "clippedsoftmax1(-.025:1)"):
        b, ga, et = tbl[sm]
        clip = not (ga == 0.0 and et == 1.0)
        out = E.attn_core_eager(q, k, v, order="bert", base=b, clip=clip, gamma=ga, eta=et, mask=torch.from_numpy(g["pad_mask"]))
        np.testing.assert_allclose(out.numpy(), g[f"bert[{sm}].ctx"], **MM)
        qs = (q * 64 ** -0.5).half().float()
        out = E.attn_core_eager(qs, k, v, order="opt", base=b, clip=clip, gamma=ga, eta=et, mask=torch.from_numpy(g["opt_mask"]))
        np.testing.assert_allclose(out.numpy(), g[f"opt[{sm}].ctx"], **MM)
    # fake-quant chain agrees with the numpy oracle (indices can flip only through matmul order)
    fq = dict(scores=(0.05, 120.0, 255.0), probs=(1 / 255.0, 0.0, 255.0), ctx=(0.01, 130.0, 255.0))
    out = E.attn_core_eager(qs, k, v, order="opt", mask=torch.from_numpy(g["opt_mask"]), fq=fq)
    want = O.attn_core(qs.numpy(), k.numpy(), v.numpy(), full_mask=g["opt_mask"], clamp_min=True, fq_scores=(0.05, 120.0),
                       fq_probs=(1 / 255.0, 0.0), fq_ctx=(0.01, 130.0))
    assert (np.abs(out.numpy() - want) > 1e-5).mean() < 2e-3
    m = E.causal_mask(2, 5)
    assert m.shape == (2, 1, 5, 5) and m[0, 0, 0, 1] == torch.finfo(torch.float32).min and m[0, 0, 1, 0] == 0


# ---- round 6: the oracle against reference outputs at the shapes the fast kernels run (inputs regenerated from tests/golden/synth.py)
def test_core_long_rows_against_reference():
    """core_attn_long.npz: OPT order at S = 512 causal (3 softmax kinds) and BERT order at 704 keys with left + right key padding."""
    from tests.golden import synth as sy

    g = load_golden("core_attn_long.npz")
    tbl = O.softmax_table()
    q, k, v = sy.long_causal_qkv()
    for sm in ("softmax1", "clippedsoftmax1(-.025:1)", "vanilla"):
        b, ga, et = tbl[sm]
        ctx, ex = O.attn_core(q, k, v, base=b, gamma=ga, eta=et, clip=not (ga == 0.0 and et == 1.0), causal=True, clamp_min=True, want=("probs",))
        np.testing.assert_allclose(ctx, g[f"opt512[{sm}].ctx"], err_msg=sm, **MM)
        np.testing.assert_allclose(ex["probs"].sum(-1), g[f"opt512[{sm}].probs_rowsum"], err_msg=sm, rtol=2e-5, atol=2e-6)
    q, k, v = sy.long_padded_qkv()
    pad = sy.key_padding(sy.LONG_PAD_B, sy.LONG_PAD_S, sy.LONG_PAD_LEFT, sy.LONG_PAD_RIGHT)
    for sm in ("softmax1", "vanilla"):
        b, ga, et = tbl[sm]
        ctx = O.attn_core(q, k, v, scale=math.sqrt(q.shape[-1]), scale_is_divisor=True, base=b, pad_mask=pad)
        np.testing.assert_allclose(ctx, g[f"bert704[{sm}].ctx"], err_msg=sm, **MM)


def _synth_state(shapes, seed, w_std):
    from tests.golden import synth as sy

    return sy.state_dict_like(shapes, seed, w_std=w_std)


def _gate_shapes(E, H, gate):
    d = E // H
    sh = {}
    if gate == "tok_linear":
        for h in range(H):
            sh[f"alpha.{h}.weight"], sh[f"alpha.{h}.bias"] = (1, d), (1,)
    elif gate == "tok_mlp":
        for h in range(H):
            sh[f"alpha.{h}.0.weight"], sh[f"alpha.{h}.0.bias"] = (d // 4, d), (d // 4,)
            sh[f"alpha.{h}.2.weight"], sh[f"alpha.{h}.2.bias"] = (1, d // 4), (1,)
    return sh


def test_modules_at_twelve_heads_against_reference():
    """bert_attn_h12.npz / opt_attn_h12.npz: E = 768, H = 12, S = 64, B = 2 - plain, clipped and gated cases."""
    from tests.golden import synth as sy

    tbl = O.softmax_table()
    E, H, B, S = sy.H12_E, sy.H12_H, sy.H12_B, sy.H12_S
    g = load_golden("bert_attn_h12.npz")
    hidden = sy.h12_hidden(6201)
    mask = sy.key_padding(B, S, [0, 0], [0, 15]).reshape(B, 1, 1, S)
    for c in json.loads(str(g["meta_json"])):
        shapes = {f"{n}.{p_}": ((E, E) if p_ == "weight" else (E,)) for n in ("query", "key", "value") for p_ in ("weight", "bias")}
        shapes.update(_gate_shapes(E, H, c["gate"]))
        sd = _synth_state(shapes, c["seed"], c["w_std"])
        b, ga, et = tbl[c["softmax"]]
        ctx = O.bert_self_attention(sd, hidden, H, mask=mask, base=b, gamma=ga, eta=et, clip=not (ga == 0.0 and et == 1.0))
        np.testing.assert_allclose(ctx, g[f"[{c['softmax']}|{c['gate']}].ctx"], err_msg=str(c), rtol=2e-5, atol=5e-6)
    g = load_golden("opt_attn_h12.npz")
    hidden = sy.h12_hidden(6202)
    mask = sy.opt_decoder_mask(B, S, [S, 50])
    for c in json.loads(str(g["meta_json"])):
        shapes = {f"{n}.{p_}": ((E, E) if p_ == "weight" else (E,)) for n in ("q_proj", "k_proj", "v_proj", "out_proj") for p_ in ("weight", "bias")}
        shapes.update(_gate_shapes(E, H, c["gate"]))
        sd = _synth_state(shapes, c["seed"], c["w_std"])
        b, ga, et = tbl[c["softmax"]]
        out = O.opt_attention(sd, hidden, H, mask=mask, base=b, gamma=ga, eta=et, clip=not (ga == 0.0 and et == 1.0))
        np.testing.assert_allclose(out, g[f"[{c['softmax']}|{c['gate']}].out"], err_msg=str(c), rtol=2e-5, atol=2e-5)


def test_vit_small_size_against_reference():
    """vit_attn_s16.npz: ViT-S/16's attention (C = 384, 6 heads, 197 tokens) - the un-gated cases through the oracle's core
    (vit_attention.py:54-75,202-269: fused qkv Linear, q @ k^T * scale, softmax_fn, @ v, proj)."""
    from tests.golden import synth as sy

    g = load_golden("vit_attn_s16.npz")
    tbl = O.softmax_table()
    B, N, C, H = sy.VIT_B, sy.VIT_N, sy.VIT_C, sy.VIT_H
    d = C // H
    x = sy.vit_tokens(6301)
    for c in json.loads(str(g["meta_json"])):
        if c["gate"] != "nogate":
            continue
        sd = _synth_state({"qkv.weight": (3 * C, C), "qkv.bias": (3 * C,), "proj.weight": (C, C), "proj.bias": (C,)}, c["seed"], c["w_std"])
        qkv = O.linear(x, sd["qkv.weight"], sd["qkv.bias"]).reshape(B, N, 3, H, d).transpose(2, 0, 3, 1, 4)
        b, ga, et = tbl[c["softmax"]]
        ctx = O.attn_core(qkv[0], qkv[1], qkv[2], scale=d ** -0.5, base=b, gamma=ga, eta=et, clip=not (ga == 0.0 and et == 1.0))
        out = O.linear(O.merge_heads(ctx), sd["proj.weight"], sd["proj.bias"])
        np.testing.assert_allclose(out, g[f"[{c['softmax']}|{c['gate']}].out"], err_msg=str(c), rtol=2e-5, atol=2e-5)
