"""Drop-in modules on the GPU against module-level I/O captured from the reference (tests/golden/*.npz).
Modules run in fp32 here (the reference captured fp32): the kernel rounds q/k/v to fp16 for the MFMA operands, so
the tolerance is the fp16 contract (1e-3 + 1e-3*|ref|, tests/test_attn_gpu.py) widened to 2e-3 for the out_proj that
follows in OPT/ViT.  `-m gpu`."""
import json
import os

import numpy as np
import pytest

from tests.conftest import load_golden
from tests.test_host_cpu import Cfg, gate_kwargs

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL = dict(atol=5e-4, rtol=5e-4)  # fp32 modules through the fp32-storage kernels (operand pairs): the kernels' own 5e-4


@pytest.fixture(scope="module")
def oa():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import outeffhop_amd

    return outeffhop_amd


def _close(got, want, msg, tol=TOL):
    got = got.detach().float().cpu().numpy()
    err = np.abs(got - want)
    lim = tol["atol"] + tol["rtol"] * np.abs(want)
    assert np.isfinite(got).all() and (err <= lim).all(), f"{msg}: max err {err.max():.3e}"


def _sd(g, name, base_prefix="w."):
    sd = {k[len(base_prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(base_prefix)}
    sd.update({k[len(name) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + ".w.")})
    return sd


def test_bert_module_all_cases(oa):
    g = load_golden("bert_attn_fp.npz")
    hidden = torch.from_numpy(g["hidden"]).cuda()
    mask = torch.from_numpy(g["mask"]).cuda()
    for cj in g["cases_json"]:
        c = json.loads(str(cj))
        m = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING[c["softmax"]], **gate_kwargs(c["gate"]))
        m.load_state_dict(_sd(g, c["name"]), strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            (ctx,) = m(hidden, attention_mask=mask)                          # fused kernel
            ctx2, probs = m(hidden, attention_mask=mask, output_attentions=True)  # observable path
            (ctx3,) = m(hidden)
        _close(ctx, g[f"{c['name']}.ctx"], c["name"] + " fused")
        _close(ctx2, g[f"{c['name']}.ctx"], c["name"] + " unfused", dict(atol=2e-5, rtol=1e-4))
        _close(probs, g[f"{c['name']}.probs"], c["name"] + " probs", dict(atol=2e-6, rtol=1e-4))
        _close(ctx3, g[f"{c['name']}.ctx_nomask"], c["name"] + " nomask")
        if f"{c['name']}.last_gate_avg_prob" in g.files:
            _close(m.last_gate_avg_prob, g[f"{c['name']}.last_gate_avg_prob"], c["name"] + " gate avg", dict(atol=1e-5, rtol=1e-4))
    m = oa.BertSelfAttentionWithExtras(Cfg(), alpha=4.0, max_seq_length=32)
    m.load_state_dict(_sd(g, "_none_"), strict=True)
    with torch.no_grad():
        _close(m.cuda().eval()(hidden, attention_mask=mask)[0], g["alpha4.ctx"], "alpha=4")
    # fp16 module (BASELINE config 2 dtype): fp32 math on fp16 data vs the reference's pure-fp16 eager result
    m = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"])
    m.load_state_dict(_sd(g, "_none_"), strict=True)
    m = m.cuda().half().eval()
    hm = torch.zeros_like(mask, dtype=torch.float16).masked_fill_(mask != 0, torch.finfo(torch.float16).min)
    with torch.no_grad():
        out = m(hidden.half(), attention_mask=hm)[0]
    assert out.dtype == torch.float16
    _close(out, g["half.ctx"].astype(np.float32), "half", dict(atol=4e-3, rtol=4e-3))
    # a forward hook on a tap forces (and sees) the materialised scores
    m = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"])
    m.load_state_dict(_sd(g, "_none_"), strict=True)
    m = m.cuda().eval()
    seen = []
    m.attn_probs_before_dropout.register_forward_hook(lambda mod, i, o: seen.append(o))
    with torch.no_grad():
        m(hidden, attention_mask=mask)
    _close(seen[0], g["sm[softmax1].probs"], "hooked probs", dict(atol=2e-6, rtol=1e-4))


def test_opt_module_all_cases(oa):
    g = load_golden("opt_attn_fp.npz")
    hidden = torch.from_numpy(g["hidden"]).cuda()
    mask = torch.from_numpy(g["mask"]).cuda()
    for cj in g["cases_json"]:
        c = json.loads(str(cj))
        m = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING[c["softmax"]], **gate_kwargs(c["gate"]))
        m.load_state_dict(_sd(g, c["name"]), strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            out, w, past = m(hidden, attention_mask=mask)
            out2, w2, _ = m(hidden, attention_mask=mask, output_attentions=True)
            out3, _, _ = m(hidden)
        assert w is None and past[0].shape == (2, 2, 32, 64)
        _close(out, g[f"{c['name']}.out"], c["name"] + " fused")
        _close(out2, g[f"{c['name']}.out"], c["name"] + " unfused", dict(atol=3e-5, rtol=1e-4))
        _close(w2, g[f"{c['name']}.probs"].reshape(w2.shape), c["name"] + " probs", dict(atol=2e-6, rtol=1e-4))
        _close(out3, g[f"{c['name']}.out_nomask"], c["name"] + " nomask")
    _close(past[0], g["past_k"], "past k", dict(atol=1e-5, rtol=1e-5))
    # causal+padding mask is recognised and replaced by the analytic causal flag + padding vector
    from outeffhop_amd.attention import classify_causal

    ok, pad = classify_causal(mask)
    assert ok and pad is not None and pad.shape == (2, 32)
    ok, pad = classify_causal(mask[:1].contiguous())
    assert ok and pad is None
    assert classify_causal(torch.zeros_like(mask)) == (False, None)
    # fp16 module + softmax1 runs (TypeError in the reference)
    m = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"])
    m.load_state_dict(_sd(g, "sm[softmax1]"), strict=True)
    m = m.cuda().half().eval()
    hm = torch.zeros_like(mask, dtype=torch.float16).masked_fill_(mask != 0, torch.finfo(torch.float16).min)
    with torch.no_grad():
        out = m(hidden.half(), attention_mask=hm)[0]
    _close(out, g["sm[softmax1].out"], "fp16 softmax1", dict(atol=6e-3, rtol=6e-3))
    m = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, alpha=12.0, max_seq_length=32, attn_softmax="softmax1")
    m.load_state_dict(_sd(g, "_none_"), strict=True)
    assert str(g["alpha12.softmax_fn_name"]) == "clipped_softmax1"
    with torch.no_grad():
        _close(m.cuda().eval()(hidden, attention_mask=mask)[0], g["alpha12.out"], "alpha=12")


def test_vit_module(oa):
    g = load_golden("vit_attn_fp.npz")
    x = torch.from_numpy(g["x"]).cuda()
    for cj in g["cases_json"]:
        c = json.loads(str(cj))
        m = oa.ViTSelfAttentionWithExtras(128, num_heads=2, qkv_bias=True, softmax_fn=oa.SOFTMAX_MAPPING[c["softmax"]], **gate_kwargs(c["gate"]))
        m.load_state_dict(_sd(g, c["name"], base_prefix="__none__"), strict=True)
        with torch.no_grad():
            _close(m.cuda().eval()(x), g[f"{c['name']}.out"], c["name"])


def test_stanhop_association_and_hopfield(oa):
    g = load_golden("stanhop_assoc.npz")
    q, k, v = (torch.from_numpy(g[n]).cuda() for n in ("q", "k", "v"))
    for mode in ("softmax1", "softmax", "clip"):
        out = oa.Association(mode=mode).eval()(q, k, v)
        assert out.shape == (3, 7, 4, 16) and out.is_contiguous()
        _close(out, g[f"assoc[{mode}]"], mode)
    _close(oa.Association(mode="softmax1", scale=0.37).eval()(q, k, v), g["assoc[softmax1,scale=0.37]"], "scale")
    assert str(g["clip_softmax1_ctor_error"]) == "TypeError"  # reference bug; ours constructs and runs
    oa.Association(mode="clip_softmax1").eval()(q, k, v)
    hop = oa.Hopfield(64, 4, mode="softmax1")
    hop.load_state_dict({kk[6:]: torch.from_numpy(g[kk]) for kk in g.files if kk.startswith("hop.w.")}, strict=True)
    hop = hop.cuda().eval()
    x, y = torch.from_numpy(g["hop.x"]).cuda(), torch.from_numpy(g["hop.y"]).cuda()
    with torch.no_grad():
        _close(hop(x, x, x), g["hop.self"], "hopfield self")
        _close(hop(x, y, y), g["hop.cross"], "hopfield cross (L != S)")
    pool = oa.HopfieldPooling(64, 4, num_pattern=3, mode="softmax1")
    pool.load_state_dict({kk[7:]: torch.from_numpy(g[kk]) for kk in g.files if kk.startswith("pool.w.")}, strict=True)
    with torch.no_grad():
        _close(pool.cuda().eval()(x), g["pool.out"], "pooling")


def test_theory_verification_config1(oa):
    """BASELINE config 1: single Hopfield layer, softmax1, B=4 S=64 d=32 (n_heads=1)."""
    g = load_golden("theory_hopfield_cfg1.npz")
    q, k, v = (torch.from_numpy(g[n]).cuda() for n in ("assoc_q", "assoc_k", "assoc_v"))
    out = oa.Association(mode="softmax1").eval()(q, k, v)
    _close(out, g["assoc_out_softmax1"], "cfg1 association")
    w = {kk[2:]: torch.from_numpy(g[kk]).cuda() for kk in g.files if kk.startswith("w.")}
    full = torch.nn.functional.linear(out.reshape(4, 64, -1), w["out_projection.weight"], w["out_projection.bias"])
    _close(full, g["out_softmax1"], "cfg1 layer output")


def _load_synth(mod, seed, w_std):
    from tests.golden import synth as sy

    shapes = {k_: tuple(v_.shape) for k_, v_ in mod.state_dict().items()}
    mod.load_state_dict({k_: torch.from_numpy(v_) for k_, v_ in sy.state_dict_like(shapes, seed, w_std=w_std).items()}, strict=True)
    return mod


def test_modules_at_twelve_heads_against_the_reference(oa):
    """Round 6 (VERDICT r5 next #2b): BERT-base / OPT-125m WIDTH (E = 768, H = 12, d = 64; S = 64, B = 2) - the 12-head strides the
    BASELINE configs run with - against module outputs captured from the reference (bert_attn_h12.npz / opt_attn_h12.npz); weights and
    inputs regenerated from tests/golden/synth.py.  fp32 modules (what the reference captured) and fp16 modules (fp16 kernels, the in-kernel
    gate predictor) on the same weights."""
    from tests.golden import synth as sy

    class Cfg12(Cfg):
        hidden_size = 768
        num_attention_heads = 12
        max_position_embeddings = 512

    B, S, E, H = sy.H12_B, sy.H12_S, sy.H12_E, sy.H12_H
    fmin16 = torch.finfo(torch.float16).min
    g = load_golden("bert_attn_h12.npz")
    hidden = torch.from_numpy(sy.h12_hidden(6201)).cuda()
    mask = torch.from_numpy(sy.key_padding(B, S, [0, 0], [0, 15])).view(B, 1, 1, S).cuda()
    for c in json.loads(str(g["meta_json"])):
        want = g[f"[{c['softmax']}|{c['gate']}].ctx"]
        m = _load_synth(oa.BertSelfAttentionWithExtras(Cfg12(), softmax_fn=oa.SOFTMAX_MAPPING[c["softmax"]], **gate_kwargs(c["gate"])), c["seed"], c["w_std"]).cuda().eval()
        with torch.no_grad():
            got = m(hidden, attention_mask=mask)[0]
            _close(got, want, f"bert h12 {c} fp32")
            if c["gate"] != "nogate":
                _close(m.last_gate_avg_prob, g[f"[{c['softmax']}|{c['gate']}].last_gate_avg_prob"], f"bert h12 {c} gate avg", dict(atol=1e-4, rtol=1e-4))
            got16 = m.half()(hidden.half(), attention_mask=mask.half().clamp(min=fmin16))[0]
        # fp16 module: fp16 weights and hidden states (2^-11 relative each, through a 768-term dot product) on top of the kernel's contract
        _close(got16, want, f"bert h12 {c} fp16", dict(atol=4e-3, rtol=4e-3))   # (measured 1.3e-3 ... 2.2e-3)
        print(f"bert h12 {c['softmax']}|{c['gate']}: fp32 module max err {float(np.abs(got.cpu().numpy() - want).max()):.2e}, fp16 module {float(np.abs(got16.float().cpu().numpy() - want).max()):.2e}")
    g = load_golden("opt_attn_h12.npz")
    hidden = torch.from_numpy(sy.h12_hidden(6202)).cuda()
    mask = torch.from_numpy(sy.opt_decoder_mask(B, S, [S, 50])).cuda()
    for c in json.loads(str(g["meta_json"])):
        want = g[f"[{c['softmax']}|{c['gate']}].out"]
        m = _load_synth(oa.OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING[c["softmax"]], **gate_kwargs(c["gate"])), c["seed"], c["w_std"]).cuda().eval()
        with torch.no_grad():
            got = m(hidden, attention_mask=mask)[0]
            _close(got, want, f"opt h12 {c} fp32", dict(atol=2e-3, rtol=2e-3))
            got16 = m.half()(hidden.half(), attention_mask=mask.half().clamp(min=fmin16))[0]
        _close(got16, want, f"opt h12 {c} fp16", dict(atol=4e-3, rtol=4e-3))    # (measured 1.2e-3 ... 2.2e-3)
        print(f"opt h12 {c['softmax']}|{c['gate']}: fp32 module max err {float(np.abs(got.cpu().numpy() - want).max()):.2e}, fp16 module {float(np.abs(got16.float().cpu().numpy() - want).max()):.2e}")


# (measured on MI355X, round 6: attention / projection quantisers 4e-7 and 6e-5; out_proj's OUTPUT quantiser 3e-5 and 4e-3 - its
# input is the quantised context, where one index flip in 1e5 among the ~60 largest outputs moves the 99.999th percentile)
# eval on the reference's grids: 2.0e-3 of 41 440 sampled outputs one output-grid step off (never more) - every output sums 768 context indices
# x weights, and the share of context indices that differ from the reference's by one step is 1.5e-5 (test_attn_gpu.py: cfg4 full size)
def test_vit_small_size_against_the_reference(oa):
    """ViT-S/16's attention at its own size (C = 384, 6 heads of 64, N = 197 tokens: three full key tiles + a ragged one of 5 keys - the tile the
    one-pass kernel now specialises) against module outputs captured from the reference (vit_attn_s16.npz); fp32 and fp16 modules."""
    from tests.golden import synth as sy

    g = load_golden("vit_attn_s16.npz")
    x = torch.from_numpy(sy.vit_tokens(6301)).cuda()
    for c in json.loads(str(g["meta_json"])):
        want = g[f"[{c['softmax']}|{c['gate']}].out"]
        m = _load_synth(oa.ViTSelfAttentionWithExtras(sy.VIT_C, num_heads=sy.VIT_H, qkv_bias=True, softmax_fn=oa.SOFTMAX_MAPPING[c["softmax"]], **gate_kwargs(c["gate"])),
                        c["seed"], c["w_std"]).cuda().eval()
        with torch.no_grad():
            got = m(x)
            _close(got, want, f"vit-s {c} fp32", dict(atol=5e-4, rtol=5e-4))   # (measured 7e-5 ... 1.1e-4)
            got16 = m.half()(x.half())
        _close(got16, want, f"vit-s {c} fp16", dict(atol=2e-3, rtol=2e-3))   # (measured 3e-4 ... 6e-4)
        print(f"vit-s {c['softmax']}|{c['gate']}: fp32 module max err {float(np.abs(got.cpu().numpy() - want).max()):.2e}, fp16 module {float(np.abs(got16.float().cpu().numpy() - want).max()):.2e}")


def test_bert_int8_calibration_and_eval_at_full_size_against_the_reference(oa):
    """The BERT twin of the cfg4 test: the reference's QuantizedBertSelfAttentionWithExtras at BERT-base size (E = 768, H = 12, S = 128, B = 32, key padding,
    softmax1; quantized_bert.py:268-440) - quantiser scalars after every one of 4 calibration batches, then (on the reference's grids) sampled outputs and the
    index histograms of the three attention quantisers for the evaluation batch (tests/golden/bert_int8_calib.npz)."""
    from outeffhop_amd import ops
    from tests.golden import synth as sy

    class Cfg12(Cfg):
        hidden_size = 768
        num_attention_heads = 12
        max_position_embeddings = 512

    g = load_golden("bert_int8_calib.npz")
    B, S, E, H = sy.BI8_B, sy.BI8_S, sy.BI8_E, sy.BI8_H
    dev = torch.device("cuda:0")
    org = _load_synth(oa.BertSelfAttentionWithExtras(Cfg12(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"]), sy.BI8_WEIGHT_SEED, 0.05)
    qm = oa.QuantizedBertSelfAttentionWithExtras(org.to(dev), **_qparams(oa)).to(dev).eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    lens = sy.bi8_lengths()
    padv = torch.from_numpy(sy.key_padding(B, S, [0] * B, [S - n for n in lens])).to(dev)
    mask = padv.view(B, 1, 1, S)
    names = [n for n, m_ in qm.named_modules() if hasattr(m_, "quantizer") and not n.endswith("range_estimator")]
    with torch.no_grad():
        for i, seed in enumerate(sy.BI8_CALIB_SEEDS):
            qm(torch.from_numpy(sy.bi8_hidden(seed)).to(dev), attention_mask=mask)
            wd = wz = 0.0
            for n in names:
                qz = qm.get_submodule(n).quantizer
                key = f"after{i + 1}.q.{n}.delta"
                if key in g.files and getattr(qz, "_delta", None) is not None:
                    wd = max(wd, abs(float(qz._delta) - float(g[key])) / float(g[key]))
                    zk = f"after{i + 1}.q.{n}.zero_float"
                    if zk in g.files:
                        wz = max(wz, abs(float(qz._zero_float) - float(g[zk])))
            print(f"bert int8 calibration batch {i + 1}: worst delta rel err {wd:.2e}, worst zero_float abs err {wz:.2e}")
            assert wd <= CFG4_LIMITS["delta"] and wz <= CFG4_LIMITS["zero"], (i, wd, wz)
        qm.fix_ranges()
        for n in names:   # evaluation on the reference's grids (see the cfg4 test)
            qz = qm.get_submodule(n).quantizer
            if f"final.q.{n}.delta" in g.files and getattr(qz, "_delta", None) is not None:
                qz._delta.copy_(torch.as_tensor(float(g[f"final.q.{n}.delta"]), dtype=qz._delta.dtype))
                if f"final.q.{n}.zero_float" in g.files:
                    qz._zero_float.copy_(torch.as_tensor(float(g[f"final.q.{n}.zero_float"]), dtype=qz._zero_float.dtype))
        x = torch.from_numpy(sy.bi8_hidden(sy.BI8_EVAL_SEED)).to(dev)
        out = qm(x, attention_mask=mask)[0]
        step = float(g["final.q.context_act_quantizer.activation_quantizer.delta"])
        err = np.abs(out[::2, ::5, ::13].float().cpu().numpy() - g["eval.out_sample"])
        off, steps = float((err > 0.5 * step).mean()), float(err.max() / step)
        print(f"bert int8 eval: sampled outputs > half a step off {off:.2e}, max error {steps:.2f} steps; |out| max {float(out.abs().max()):.4f} (reference {float(g['eval.out_absmax']):.4f})")
        assert steps <= CFG4_LIMITS["steps"] and off <= CFG4_LIMITS["off"]
        d = E // H
        heads = lambda t: t.view(B, S, H, d).permute(0, 2, 1, 3)  # noqa: E731
        q, k, v = heads(qm.query(x)), heads(qm.key(x)), heads(qm.value(x))
        dumps = [torch.zeros((B, H, S, S), dtype=torch.uint8, device=dev), torch.zeros((B, H, S, S), dtype=torch.uint8, device=dev),
                 torch.zeros((B, H, S, d), dtype=torch.uint8, device=dev)]
        FQ = ops.FakeQuantSpec.from_delta
        trio = [getattr(qm, n).activation_quantizer.quantizer for n in ("attn_scores_act_quantizer", "attn_probs_act_quantizer", "context_act_quantizer")]
        fq = ops.AttnFakeQuant(*(FQ(float(z._delta), float(z._zero_float), dump=t_) for z, t_ in zip(trio, dumps)), ctx_before_gate=False)
        ops.attn_fwd(q, k, v, fq=fq, scale_div=8.0, key_pad_mask=padv, mask_min=float(np.finfo(np.float32).min))
        for name, t_ in zip(("scores", "probs", "ctx"), dumps):
            h = torch.bincount(t_.flatten().to(torch.int64), minlength=256).cpu().numpy()
            ref = g[f"eval.hist.{name}"]
            moved = float(np.abs(h - ref).sum()) / 2.0 / float(ref.sum())
            print(f"bert int8 eval {name} index histogram: share of indices in another bin than the reference's {moved:.2e}")
            assert h.sum() == ref.sum() and moved <= CFG4_LIMITS["hist"], (name, moved)


CFG4_LIMITS = dict(delta=2e-6, zero=2e-4, delta_out=1e-4, zero_out=1e-2, hist=1e-4, steps=1.05, off=4e-3)   # hist: measured 7e-6 / 2e-6 / 3.6e-5 (scores / probs / ctx); the suite's FLIP_RATE


def test_cfg4_calibration_and_eval_at_full_size_against_the_reference(oa):
    """Round 6 (VERDICT r5 next #2c; BASELINE config 4): the reference's QuantizedOPTAttentionWithExtras at OPT-125m size
    (E = 768, H = 12, S = 512, B = 16) calibrated over 4 batches (percentile 99.999, EMA 0.9: range_estimators.py:83-106) - its quantiser
    scalars after EVERY batch, then for the evaluation batch the index histograms of the three attention quantisers and sampled outputs
    (tests/golden/cfg4_calib.npz: scalars and 3 x 256 integers).  Here: the same flow through `oeh_attn_calibrate` + `oeh_percentile_ema`
    (no S x S tensor), fix_ranges, the fused INT8 forward; the histograms from the index dumps of `oeh_attn_fwd` on the module's own
    quantised q / k / v.  Limits (measured values are printed; CFG4_LIMITS): the calibrated delta of the three attention quantisers and of the
    q / k / v projections' output quantisers within 2e-6 relative (measured 4e-7) and their zero_float within 2e-4 absolute after EACH batch;
    out_proj's output quantiser - downstream of the quantised context - within 1e-4 / 1e-2; index histograms equal up to a 1e-4 share of moved
    indices (measured 7e-6 / 2e-6 / 3.6e-5); with the reference's grids loaded, sampled outputs never more than one output-grid step off and at
    most 4e-3 of them one step off (measured 2.0e-3)."""
    from outeffhop_amd import ops
    from tests.golden import synth as sy

    g = load_golden("cfg4_calib.npz")
    B, S, E, H = sy.CFG4_B, sy.CFG4_S, sy.CFG4_E, sy.CFG4_H
    dev = torch.device("cuda:0")
    org = _load_synth(oa.OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"]), sy.CFG4_WEIGHT_SEED, 0.05)
    qm = oa.QuantizedOPTAttentionWithExtras(org.to(dev), **_qparams(oa)).to(dev).eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    mask = torch.from_numpy(sy.opt_decoder_mask(B, S, [S] * B)).to(dev)
    names = [n for n, m_ in qm.named_modules() if hasattr(m_, "quantizer") and not n.endswith("range_estimator")]

    def scalars(prefix):
        """worst relative error of a calibrated delta / absolute error of a zero_float: (attention + q/k/v projection quantisers, out_proj's output quantiser)"""
        worst = {False: [0.0, 0.0], True: [0.0, 0.0]}
        detail = []
        for n in names:
            qz = qm.get_submodule(n).quantizer
            key = f"{prefix}.q.{n}.delta"
            if key not in g.files or getattr(qz, "_delta", None) is None:
                continue
            ref_d = float(g[key])
            w_ = worst[n.startswith("out_proj.activation")]
            w_[0] = max(w_[0], abs(float(qz._delta) - ref_d) / ref_d)
            zk = f"{prefix}.q.{n}.zero_float"
            if zk in g.files:
                w_[1] = max(w_[1], abs(float(qz._zero_float) - float(g[zk])))
            detail.append(f"{n.replace('.activation_quantizer', '').replace('_act_quantizer', '')}: {abs(float(qz._delta) - ref_d) / ref_d:.1e}")
        if os.environ.get("OEH_TEST_VERBOSE"):
            print("   ", prefix, "; ".join(detail))
        return worst[False], worst[True]

    with torch.no_grad():
        for i, seed in enumerate(sy.CFG4_CALIB_SEEDS):
            qm(torch.from_numpy(sy.cfg4_hidden(seed)).to(dev), attention_mask=mask)
            (wd, wz), (od, oz) = scalars(f"after{i + 1}")
            print(f"cfg4 calibration batch {i + 1}: worst delta rel err {wd:.2e}, worst zero_float abs err {wz:.2e}; out_proj's output quantiser {od:.2e} / {oz:.2e}")
            assert wd <= CFG4_LIMITS["delta"] and wz <= CFG4_LIMITS["zero"] and od <= CFG4_LIMITS["delta_out"] and oz <= CFG4_LIMITS["zero_out"], (i, wd, wz, od, oz)
        assert qm.__dict__.get("_fused_calib_calls", 0) == len(sy.CFG4_CALIB_SEEDS), "calibration materialised the score tensors"
        qm.fix_ranges()
        # the evaluation runs on the REFERENCE's calibrated grids (the fixture's `final.*` scalars written into the quantisers): a 3e-5 shift of the
        # output grid alone would move ~1 % of the outputs that sit near a rounding boundary, and hide what the eval kernels themselves do
        for n in names:
            qz = qm.get_submodule(n).quantizer
            if f"final.q.{n}.delta" in g.files and getattr(qz, "_delta", None) is not None:
                qz._delta.copy_(torch.as_tensor(float(g[f"final.q.{n}.delta"]), dtype=qz._delta.dtype))
                if f"final.q.{n}.zero_float" in g.files:
                    qz._zero_float.copy_(torch.as_tensor(float(g[f"final.q.{n}.zero_float"]), dtype=qz._zero_float.dtype))
        x = torch.from_numpy(sy.cfg4_hidden(sy.CFG4_EVAL_SEED)).to(dev)
        out = qm(x, attention_mask=mask)[0]
        assert qm.__dict__.get("_i8_calls", 0) >= 1, "the integer-matrix-core path did not run"
        # ---- sampled outputs, in steps of the output quantiser's grid
        step = float(g["final.q.out_proj.activation_quantizer.delta"])
        err = np.abs(out[::2, ::7, ::11].float().cpu().numpy() - g["eval.out_sample"])
        off, steps = float((err > 0.5 * step).mean()), float(err.max() / step)
        am, mean = float(out.abs().max()), float(out.abs().mean())
        print(f"cfg4 eval: sampled outputs > half a step off {off:.2e}, max error {steps:.2f} steps; |out| max {am:.4f} (reference {float(g['eval.out_absmax']):.4f}), "
              f"mean {mean:.6f} (reference {float(g['eval.out_absmean']):.6f})")
        assert steps <= CFG4_LIMITS["steps"] and off <= max(CFG4_LIMITS["off"], 2.0 / err.size)
        assert abs(am - float(g["eval.out_absmax"])) <= 1.05 * step and abs(mean - float(g["eval.out_absmean"])) <= 1e-3 * float(g["eval.out_absmean"])
        # ---- index histograms of the three attention quantisers: oeh_attn_fwd with index dumps on the module's own quantised q / k / v
        d = E // H
        heads = lambda t: t.view(B, S, H, d).permute(0, 2, 1, 3)  # noqa: E731
        q, k, v = heads(qm.q_proj(x) * qm.scaling), heads(qm.k_proj(x)), heads(qm.v_proj(x))
        dumps = [torch.zeros((B, H, S, S), dtype=torch.uint8, device=dev), torch.zeros((B, H, S, S), dtype=torch.uint8, device=dev),
                 torch.zeros((B, H, S, d), dtype=torch.uint8, device=dev)]
        FQ = ops.FakeQuantSpec.from_delta
        trio = [getattr(qm, n).activation_quantizer.quantizer for n in ("attn_scores_act_quantizer", "attn_probs_act_quantizer", "context_act_quantizer")]
        fq = ops.AttnFakeQuant(*(FQ(float(z._delta), float(z._zero_float), dump=t_) for z, t_ in zip(trio, dumps)), ctx_before_gate=True)
        ops.attn_fwd(q, k, v, fq=fq, causal=True, clamp_min=True, mask_min=float(np.finfo(np.float32).min))
        for name, t_ in zip(("scores", "probs", "ctx"), dumps):
            h = torch.bincount(t_.flatten().to(torch.int64), minlength=256).cpu().numpy()
            ref = g[f"eval.hist.{name}"]
            assert h.sum() == ref.sum()
            moved = float(np.abs(h - ref).sum()) / 2.0 / float(ref.sum())
            print(f"cfg4 eval {name} index histogram: share of indices in another bin than the reference's {moved:.2e} (bins used {int((ref > 0).sum())}, "
                  f"saturated low / high {int(ref[0])} / {int(ref[255])})")
            assert moved <= CFG4_LIMITS["hist"], (name, moved)


def _qparams(oa):
    cfg = oa.get_quant_config()
    cfg.act_quant.options = dict(percentile=99.999)
    return {**oa.val_qparams(cfg), "quant_dict": {}}


# Measured on MI355X (round 3; the prints below, against the reference's captured module I/O in tests/golden/int8_attn.npz), all
# eight cases of either family, `default` and `plain` alike: calibrated deltas within 4e-7 relative of the reference's, EVERY
# output on the reference's grid point (0 outputs off, max error 0.00 steps).  The bounds leave room for one-in-a-thousand
# single steps at rounding boundaries on other boxes / library versions, nothing more (round 2 accepted 5 % / 15 % and 2-3 steps).
INT8_MODULE_LIMITS = {
    # family: (relative error of a calibrated delta, share of outputs more than half an output-grid step off, largest error in steps)
    "bert": (2e-6, 1e-3, 1.05),
    "opt": (2e-6, 1e-3, 1.05),
}


@pytest.mark.parametrize("accel", ["default", "plain"])
@pytest.mark.parametrize("fam", ["bert", "opt"])
def test_int8_modules_calibrate_fix_eval(oa, fam, accel):
    """The reference's INT8 validate flow on the module (quantized_bert.py:221-440, quantized_opt.py:54-274): 4 calibration
    batches in estimate_ranges state (device-side percentile + EMA), fix_ranges, then the fused kernels; calibrated ranges and
    the eval output against the reference's captured ones.  `default`: what ships - QuantLinear as one fp16 GEMM on operand
    pairs (PAIR_GEMM) and, for OPT, the integer-matrix-core attention (INT8_STORAGE); `plain`: torch's fp32 GEMMs and the
    fake-quant kernels on float values.  The measured numbers are printed and bounded by about twice
    their values: the module's output passes through three (BERT) / five (OPT: + out_proj and its output quantiser) rounding
    stages after the first quantiser, each of which can move a value that sits on a rounding boundary by one grid step."""
    from outeffhop_amd import quantization as Q

    g = load_golden("int8_attn.npz")
    calib = [torch.from_numpy(g[f"calib{i}"]).cuda() for i in range(4)]
    evalx = torch.from_numpy(g["eval"]).cuda()
    bmask, omask = torch.from_numpy(g["bert_mask"]).cuda(), torch.from_numpy(g["opt_mask"]).cuda()
    keep = (Q.PAIR_GEMM, Q.INT8_STORAGE)
    if accel == "plain":
        Q.PAIR_GEMM, Q.INT8_STORAGE = False, False
    lim_d, lim_off, lim_steps = INT8_MODULE_LIMITS[fam]
    try:
        for meta in json.loads(str(g["meta_json"])):
            pre = f"{fam}{meta['tag']}"
            sd = {k[len(pre) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + ".w.")}
            if fam == "bert":
                org = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING[meta["softmax"]], **gate_kwargs(meta["gate"]))
                org.load_state_dict(sd, strict=True)
                qm = oa.QuantizedBertSelfAttentionWithExtras(org.cuda(), **_qparams(oa)).cuda().eval()
                fwd = lambda x: qm(x, attention_mask=bmask)[0]  # noqa: E731
            else:
                org = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING[meta["softmax"]], **gate_kwargs(meta["gate"]))
                org.load_state_dict(sd, strict=True)
                qm = oa.QuantizedOPTAttentionWithExtras(org.cuda(), **_qparams(oa)).cuda().eval()
                fwd = lambda x: qm(x, attention_mask=omask)[0]  # noqa: E731
            qm.set_quant_state(weight_quant=True, act_quant=True)
            with torch.no_grad():
                for c in calib:
                    fwd(c)
                # estimate_ranges ran without the (B,H,S,S) tensors (oeh_attn_calibrate), in either configuration
                assert qm.__dict__.get("_fused_calib_calls", 0) == len(calib), "calibration materialised the score tensors"
                qm.fix_ranges()
                assert qm._fq(fam == "opt") is not None
                out = fwd(evalx)
                if accel == "default" and qm.__dict__.get("_i8_calls", 0):
                    # (round 4) the integer core's operands came from the ONE-kernel projections (oeh_proj_quant_i8), not from library GEMMs
                    assert qm.__dict__.get("_fused_proj_calls", 0) == qm.__dict__["_i8_calls"], "the projections ran as library GEMM + quantiser passes"
            d_err = 0.0
            for name in ("attn_scores_act_quantizer", "attn_probs_act_quantizer", "context_act_quantizer"):
                qz = getattr(qm, name).activation_quantizer.quantizer
                ref_d = float(g[f"{pre}.q.{name}.activation_quantizer.delta"])
                d_err = max(d_err, abs(float(qz.delta) - ref_d) / ref_d)
            ref = g[f"{pre}.out"]
            got = out.float().cpu().numpy()
            step = float(g[f"{pre}.q.context_act_quantizer.activation_quantizer.delta"] if fam == "bert" else g[f"{pre}.q.out_proj.activation_quantizer.delta"])
            err = np.abs(got - ref)
            off, steps = float((err > 0.5 * step).mean()), float(err.max() / step)
            print(f"int8 module {pre} [{accel}]: calibrated delta rel err {d_err:.2e}, outputs > half a step off {off:.2e}, max error {steps:.2f} steps")
            assert d_err <= lim_d and off <= lim_off and steps <= lim_steps, (pre, accel, d_err, off, steps)
    finally:
        Q.PAIR_GEMM, Q.INT8_STORAGE = keep


def test_gated_modules_in_16bit_use_the_in_kernel_predictor(oa):
    """fp16 modules with the conditional per-token gate: the predictor is evaluated inside the attention kernel (BERT:
    full-row kernel, OPT causal S > 128: one-pass kernel); output and the last_gate_* bookkeeping agree with the fp32 module
    (separate gate kernel + general kernel) to fp16 accuracy."""
    from outeffhop_amd.attention import AttentionGateType as GT

    torch.manual_seed(7)
    fmin16 = torch.finfo(torch.float16).min
    # BERT, B=2, S=100, per-head MLP gate
    m32 = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"], attn_gate_type=GT.conditional_per_token,
                                         attn_gate_init=0.25, attn_gate_mlp=True).cuda().eval()
    m16 = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"], attn_gate_type=GT.conditional_per_token,
                                         attn_gate_init=0.25, attn_gate_mlp=True)
    m16.load_state_dict(m32.state_dict())
    m16 = m16.cuda().half().eval()
    E = Cfg().hidden_size
    hidden = torch.randn(2, 100, E, device="cuda").half()
    mask = torch.zeros(2, 1, 1, 100, device="cuda")
    mask[1, :, :, 77:] = torch.finfo(torch.float32).min
    with torch.no_grad():
        want = m32(hidden.float(), attention_mask=mask)[0]
        got = m16(hidden, attention_mask=mask.half().clamp(min=fmin16))[0]
    assert got.dtype == torch.float16
    _close(got, want.cpu().numpy(), "bert gated fp16", dict(atol=4e-3, rtol=4e-3))
    _close(m16.last_gate_all_probs, m32.last_gate_all_probs.cpu().numpy(), "bert gate probs", dict(atol=2e-3, rtol=2e-3))
    _close(m16.last_gate_avg_prob, m32.last_gate_avg_prob.cpu().numpy(), "bert gate avg", dict(atol=2e-3, rtol=2e-3))
    # OPT, causal, S=200 (one-pass kernel), per-head Linear gate
    o32 = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"], attn_gate_type=GT.conditional_per_token,
                                    attn_gate_init=0.25).cuda().eval()
    o16 = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"], attn_gate_type=GT.conditional_per_token,
                                    attn_gate_init=0.25)
    o16.load_state_dict(o32.state_dict())
    o16 = o16.cuda().half().eval()
    hs = torch.randn(2, 200, 128, device="cuda").half()
    cm = torch.full((200, 200), torch.finfo(torch.float32).min, device="cuda").triu(1)[None, None].expand(2, 1, 200, 200).contiguous()
    with torch.no_grad():
        want = o32(hs.float(), attention_mask=cm)[0]
        got = o16(hs, attention_mask=cm.half().clamp(min=fmin16))[0]
    _close(got, want.cpu().numpy(), "opt gated fp16", dict(atol=6e-3, rtol=6e-3))
    _close(o16.last_gate_all_probs, o32.last_gate_all_probs.cpu().numpy(), "opt gate probs", dict(atol=2e-3, rtol=2e-3))


def test_fp32_linears_as_one_fp16_gemm_on_operand_triples(oa, monkeypatch):
    """Unquantised fp32 models (the reference's validate precision): q/k/v and the output projection run as ONE fp16 GEMM with
    fp32 accumulation on operand triples (oeh_split_triples; x = xh + xl 2^-11, W = Wh + Wl 2^-11, bias inside the GEMM) - at
    least as close to the exact (float64) result as torch's fp32 GEMM, with outliers and tiny values in the input - and the
    module output is the three-Linears output to fp32 accuracy."""
    from outeffhop_amd import attention as A, ops

    torch.manual_seed(41)
    dev = torch.device("cuda:0")
    lin = torch.nn.Linear(768, 2304).to(dev)
    x = torch.randn(4, 512, 768, device=dev) * 1.5
    x[:, ::37, 11] *= 60.0       # hidden-state outliers
    x[:, 1::53, :40] *= 1e-4     # and values far below one
    with torch.no_grad():
        assert A.triple_gemm_ok(x, lin)
        got = A.linear_fp32(lin, x)
        exact = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
        e_new = float((got.double() - exact).abs().max())
        e_lib = float((lin(x).double() - exact).abs().max())
        assert e_new <= max(1.5 * e_lib, 2e-5), (e_new, e_lib)
        # the activation matrix: [xh | xh 2^-5 | xl 2^-5 | 1, 2^-5, 0 ...]
        a = ops.split_triples(x.reshape(-1, 768)).float()
        assert a.shape[1] == 3 * 768 + 8 and bool((a[:, 2304] == 1.0).all()) and bool((a[:, 2305] == 2.0 ** -5).all()) and bool((a[:, 2306:] == 0).all())
        rec = a[:, :768] + a[:, 1536:2304] * (32.0 / 2048.0)
        x2 = x.reshape(-1, 768)
        assert bool(((rec - x2).abs() <= 2.0 ** -20 * x2.abs() + 2.0 ** -30).all())
        # whole module: OPT layer, causal mask, against the three separate fp32 Linears
        m = oa.OPTAttentionWithExtras(256, 4, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
        h = torch.randn(3, 96, 256, device=dev)
        mask = _decoder_mask(3, 96, [96, 90, 96], torch.float32, dev)
        out = m(h, attention_mask=mask)[0]
        monkeypatch.setattr(A, "TRIPLE_GEMM", False)
        ref = m(h, attention_mask=mask)[0]
    assert float((out - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_module_forwards_can_be_captured_into_a_hip_graph(oa):
    """The drop-in modules launch on the current stream and never read device memory on the host (the OPT path classifies a
    mask TENSOR once - here before the capture), so a whole forward - gated BERT with key padding, OPT with a causal mask -
    can be captured with torch.cuda.graph and replayed on new data in the static input: same bits as the eager forward."""
    from outeffhop_amd.attention import AttentionGateType as GT

    torch.manual_seed(11)
    f16min = torch.finfo(torch.float16).min
    bm = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"], attn_gate_type=GT.conditional_per_token,
                                        attn_gate_init=0.25, attn_gate_mlp=True).cuda().half().eval()
    E = Cfg().hidden_size
    xb = torch.randn(4, 96, E, device="cuda").half()
    pad = torch.zeros(4, 1, 1, 96, device="cuda", dtype=torch.float16)
    pad[2, :, :, 60:] = f16min
    om = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["clippedsoftmax1(-.025:1)"]).cuda().half().eval()
    xo = torch.randn(2, 160, 128, device="cuda").half()
    cm = torch.full((160, 160), f16min, device="cuda", dtype=torch.float16).triu(1)[None, None].expand(2, 1, 160, 160).contiguous()
    with torch.no_grad():
        for fn, x in ((lambda: bm(xb, attention_mask=pad)[0], xb), (lambda: om(xo, attention_mask=cm)[0], xo)):
            fn()  # warm-up: weight caches, and the one host-side look at the OPT mask
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            x.copy_(torch.randn_like(x))  # new data in the captured input
            g.replay()
            torch.cuda.synchronize()
            got = out.clone()
            want = fn()
            assert torch.equal(got, want)


def test_fused_qkv_projection_matches_three_linears(oa):
    """one GEMM with the concatenated q/k/v weights (OPT: the q scaling folded in) feeding strided views to
    the kernel - same module output as three Linears, and the cache follows in-place weight updates."""
    import torch

    from outeffhop_amd import attention
    from outeffhop_amd.opt_attention import OPTAttentionWithExtras
    from outeffhop_amd.softmax import SOFTMAX_MAPPING

    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    fmin = torch.finfo(torch.float16).min
    m = OPTAttentionWithExtras(256, 4, is_decoder=True, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).half().eval()
    x = torch.randn(3, 70, 256, device=dev).half()
    mask = torch.full((70, 70), fmin, device=dev, dtype=torch.float16).triu(1)[None, None].expand(3, 1, 70, 70)
    with torch.no_grad():
        outs = {}
        for fused in (True, False):
            attention.FUSE_QKV = fused
            try:
                outs[fused] = m(x, attention_mask=mask)[0].float()
            finally:
                attention.FUSE_QKV = True
        assert "_oeh_qkv_cache" in m.__dict__
        assert torch.allclose(outs[True], outs[False], atol=2e-3, rtol=2e-3)
        m.q_proj.weight.mul_(0.5)  # in-place update: the version counter changes, the cache must be rebuilt
        a = m(x, attention_mask=mask)[0].float()
        attention.FUSE_QKV = False
        try:
            b = m(x, attention_mask=mask)[0].float()
        finally:
            attention.FUSE_QKV = True
        assert torch.allclose(a, b, atol=2e-3, rtol=2e-3) and not torch.allclose(a, outs[True], atol=1e-4)
    # with autograd recording the module takes the differentiable torch-op path (round 5; it raised before) - never the forward-only
    # kernels, whose output would carry no grad_fn - and a DIRECT op call with such inputs still refuses
    from outeffhop_amd._lib import OehError

    m.train()
    for p_ in m.parameters():
        p_.requires_grad_(True)
    with torch.no_grad():
        want = m(x, attention_mask=mask)[0].float()
    got = m(x, attention_mask=mask)[0]
    assert got.grad_fn is not None and torch.allclose(got.float(), want, atol=2e-3, rtol=2e-3)
    q = torch.randn(1, 2, 32, 64, device="cuda", dtype=torch.float16, requires_grad=True)
    with pytest.raises(OehError, match="forward-only"):
        oa.ops.attn_fwd(q, q, q)
    with torch.no_grad():
        assert torch.isfinite(m(x, attention_mask=mask)[0]).all()


def _decoder_mask(B, T, lens, dtype, dev, at=None):
    """HF's decoder mask (causal + right padding), as a NEW tensor; `at`: try to have the caching allocator place it at this
    address (the block the previous batch's mask has just given back) - candidates that land elsewhere are held until one fits."""
    fmin = torch.finfo(dtype).min
    m = torch.full((T, T), fmin, dtype=dtype, device=dev).triu(1)[None, None].repeat(B, 1, 1, 1)
    for b, n in enumerate(lens):
        m[b, :, :, n:] = fmin
    if at is None:
        return m.clone()
    held = []
    for _ in range(256):
        cand = torch.empty_like(m)
        if cand.data_ptr() == at:
            break
        held.append(cand)
    cand.copy_(m)
    return cand


@pytest.mark.parametrize("quantised", [False, True])
def test_opt_consecutive_batches_with_different_padding(oa, quantised):
    """VERDICT r1 weak #1 / ADVICE r1 (high): HF builds a new (B,1,T,S) decoder mask every forward and the caching allocator
    returns the same address; the causal+padding classification must follow the mask's CONTENT.  Two batches with different
    padding, the first mask freed before the second is built, fused path against the observable path (which adds the
    mask tensor itself and is pinned to the reference by test_opt_module_all_cases / test_int8_modules_calibrate_fix_eval)
    and against the reference op chain (oracle/eager_torch.py) on the module's own q, k, v."""
    from oracle import eager_torch as E
    from outeffhop_amd import attention as A

    g = load_golden("opt_attn_fp.npz")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    m = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"])
    m.load_state_dict(_sd(g, "sm[softmax1]"), strict=True)
    m = m.cuda().eval()
    B, T = 4, 96
    if quantised:
        m = oa.QuantizedOPTAttentionWithExtras(m, **_qparams(oa)).cuda().eval()
        m.set_quant_state(weight_quant=True, act_quant=True)
        with torch.no_grad():
            for i in range(2):
                m(torch.randn(B, T, 128, device=dev), attention_mask=_decoder_mask(B, T, [T] * B, torch.float32, dev))
        m.fix_ranges()
    A._causal_cache.clear()
    same_addr = 0
    addr = None
    for trial in range(6):
        lens = [T - 7 * ((trial + 2 * b) % 5) for b in range(B)]
        x = torch.randn(B, T, 128, device=dev)
        mask = _decoder_mask(B, T, lens, torch.float32, dev, at=addr)
        same_addr += int(mask.data_ptr() == addr)
        addr = mask.data_ptr()
        with torch.no_grad():
            fused = m(x, attention_mask=mask)[0]
            seen = m(x, attention_mask=mask, output_attentions=True)[0]
        tol = dict(atol=5e-4, rtol=5e-4)
        if quantised:  # outputs sit on an 8-bit grid: a step for the few elements at a rounding boundary (measured: printed)
            step = float(m.out_proj.activation_quantizer.quantizer.delta) if hasattr(m.out_proj, "activation_quantizer") else 0.05
            err = (fused - seen).abs()
            steps, off = float(err.max()) / step, float((err > 0.5 * step).float().mean())
            print(f"quantised OPT, batch {trial}: fused vs observable path max {steps:.2f} steps, {off:.2e} of the outputs more than half a step apart")
            # (the observable path's projections accumulate in the library GEMM's order, the fused path's in ops.proj_quant_i8's: a
            # projection value on a rounding boundary to within the fp32 accumulation error lands one step apart - a few in 10^5 -
            # and moves the outputs of its token by at most a step; measured: 0 ... 1.2e-3 of the outputs, never more than 1 step)
            assert steps <= 1.05 and off <= 5e-3, (trial, steps, off)
        else:
            _close(fused, seen.cpu().numpy(), f"batch {trial} fused vs observable", tol)
            # the reference op chain on the module's own projections
            with torch.no_grad():
                q = (m.q_proj(x) * m.scaling).view(B, T, 2, 64).permute(0, 2, 1, 3).cpu()
                k = m.k_proj(x).view(B, T, 2, 64).permute(0, 2, 1, 3).cpu()
                v = m.v_proj(x).view(B, T, 2, 64).permute(0, 2, 1, 3).cpu()
                ctx = E.attn_core_eager(q, k, v, order="opt", base=1, clip=False, gamma=0.0, eta=1.0, mask=mask.cpu())
                want = m.out_proj(ctx.permute(0, 2, 1, 3).reshape(B, T, 128).to(dev))
            _close(fused, want.cpu().numpy(), f"batch {trial} vs reference chain", tol)
        del mask
    assert same_addr > 0, "the allocator never reused the mask's address: the hazard was not exercised"


def test_fused_gate_falls_back_when_the_library_refuses(oa, monkeypatch):
    """`except _lib.OehError` named a module attention.py never imported - the OEH_ENOTSUP fallback to
    oeh_gate_fwd raised NameError.  Force the refusal (clipped vanilla softmax + key padding + 640 keys: the two-pass clipped form has
    no in-kernel predictor, nor has the any-shape kernel) with the probe patched to say yes."""
    from outeffhop_amd import attention as A, ops

    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    B, H, S, D = 2, 2, 640, 64
    q, k, v = (torch.randn(B, H, S, D, device=dev).half() for _ in range(3))
    hidden = torch.randn(B, S, H * D, device=dev).half()
    w1, b1 = torch.randn(H, D, device=dev) * 0.05, torch.randn(H, device=dev)
    pad = torch.zeros(B, 1, 1, S, device=dev)
    pad[1, ..., 500:] = torch.finfo(torch.float32).min
    sm = oa.SOFTMAX_MAPPING["clipped(-.003:1.003)"]
    gp = ops.GatePredictor(hidden, w1, b1, scaling=2.0, out=torch.empty(B, H, S, device=dev))
    with pytest.raises(oa._lib.OehError) as ei:  # the library does refuse this combination ...
        ops.attn_fwd(q, k, v, softmax=sm.spec, scale_div=8.0, key_pad_mask=pad, gate_mlp=gp)
    assert ei.value.code == -95
    monkeypatch.setattr(ops, "fused_gate_ok", lambda *a, **kw: True)
    got = A.attention_core(q, k, v, softmax_fn=sm, scale_div=8.0, attention_mask=pad, gate_mlp=gp)  # ... and the host falls back
    gate = ops.gate_fwd(hidden, H, w1, b1, scaling=1.0)
    want = A.attention_core(q, k, v, softmax_fn=sm, scale_div=8.0, attention_mask=pad, gate=gate * 2.0)
    assert torch.equal(got, want) and torch.equal(gp.out, gate[..., 0])


def test_quantised_opt_uses_the_int8_storage_core(oa, monkeypatch):
    """QuantizedOPTAttentionWithExtras on a purely causal mask (and without one): the q/k/v QuantLinear outputs go to the
    attention core as 8-bit indices (integer matrix cores); same module output as the fake-quant kernels on the float values
    up to rare single steps of the output grid, and the (k, v) cache it returns is the dequantised projections."""
    from outeffhop_amd import ops, quantization as Q

    torch.manual_seed(21)
    dev = torch.device("cuda:0")
    B, T, E, H = 3, 96, 256, 4
    org = oa.OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
    qm = oa.QuantizedOPTAttentionWithExtras(org, **_qparams(oa)).to(dev).eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    mask = _decoder_mask(B, T, [T] * B, torch.float32, dev)
    with torch.no_grad():
        for _ in range(3):
            qm(torch.randn(B, T, E, device=dev), attention_mask=mask)
        qm.fix_ranges()
        x = torch.randn(B, T, E, device=dev)
        monkeypatch.setattr(Q, "I8_PLAN", False)  # (this test counts the ops calls of EVERY forward: the prebuilt plan of a repeated forward bypasses them)
        calls = []
        real = ops.attn_fwd_i8
        monkeypatch.setattr(ops, "attn_fwd_i8", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        out8, w8, past8 = qm(x, attention_mask=mask)
        out8n, _, _ = qm(x)
        assert len(calls) == 2 and w8 is None
        # ... and out_proj ran on the context quantiser's integers (one 16-bit GEMM of integers, QuantLinear.linear_index); with
        # the float hand-over instead (split pass + operand-pair GEMM) the outputs agree up to rare single steps of the output grid
        assert qm.__dict__.get("_index_gemm_calls", 0) == 2
        # (round 4) ... as int8 centred indices against the int8 weights on the integer matrix cores (E % 64 == 0): exact int32 sums, so
        # the same outputs bit for bit as with the integers idx - zp carried in fp16
        assert qm.out_proj.__dict__.get("_int8_index_calls", 0) == 2
        monkeypatch.setattr(Q.QuantLinear, "int8_index_ok", lambda self, rows: False)
        out8h, _, _ = qm(x, attention_mask=mask)
        monkeypatch.undo()
        monkeypatch.setattr(Q, "I8_PLAN", False)
        monkeypatch.setattr(ops, "attn_fwd_i8", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        assert qm.out_proj.__dict__["_int8_index_calls"] == 2 and torch.equal(out8h, out8)
        calls.pop()
        qm.__dict__["_index_gemm_calls"] -= 1
        monkeypatch.setattr(Q, "INDEX_GEMM", False)
        out8v, _, _ = qm(x, attention_mask=mask)
        monkeypatch.setattr(Q, "INDEX_GEMM", True)
        assert len(calls) == 3 and qm.__dict__["_index_gemm_calls"] == 2
        calls.pop()
        stepo = float(qm.out_proj.activation_quantizer.quantizer.delta)
        erri = (out8 - out8v).abs()
        print(f"quantised OPT: out_proj on integers vs on float values: max {float(erri.max()) / stepo:.2f} steps, {float((erri > 0.5 * stepo).float().mean()):.2e} of the outputs apart")
        assert float(erri.max()) <= 1.05 * stepo and float((erri > 0.5 * stepo).float().mean()) < 1e-3
        monkeypatch.setattr(Q, "INT8_STORAGE", False)
        outf, _, pastf = qm(x, attention_mask=mask)
        outfn, _, _ = qm(x)
        assert len(calls) == 2
    step = float(qm.out_proj.activation_quantizer.quantizer.delta)
    for a, b_ in ((out8, outf), (out8n, outfn)):
        err = (a - b_).abs()
        assert float(err.max()) <= 2.05 * step and float((err > 0.5 * step).float().mean()) < 5e-3, (float(err.max()), step)
    # (the integer path's projections accumulate in their own order - ops.proj_quant_i8 - so a value on a rounding boundary may land
    # one step away from the float path's: rare, never more than one step of the projection's grid)
    for n_, lin in ((0, qm.k_proj), (1, qm.v_proj)):
        dp = (past8[n_] - pastf[n_]).abs()
        assert float(dp.max()) <= 1.001 * float(lin.activation_quantizer.quantizer.delta) and float((dp > 0).float().mean()) < 1e-4
    # a batch with padded keys (causal + finfo.min columns) runs the integer core too (its key-padding variant), same outputs
    # as the fake-quant kernels on floats up to rare single steps; a mask that is not causal + padding does not
    padded = _decoder_mask(B, T, [T, T - 5, T - 40], torch.float32, dev)
    with torch.no_grad():
        monkeypatch.setattr(Q, "INT8_STORAGE", True)
        outp8, _, _ = qm(x, attention_mask=padded)
        assert len(calls) == 3
        monkeypatch.setattr(Q, "INT8_STORAGE", False)
        outpf, _, _ = qm(x, attention_mask=padded)
        monkeypatch.setattr(Q, "INT8_STORAGE", True)
        odd = padded.clone()
        odd[0, 0, 5, 2] = -3.0  # an additive value that is neither 0 nor finfo.min
        qm(x, attention_mask=odd)
        assert len(calls) == 3
    err = (outp8 - outpf).abs()
    assert float(err.max()) <= 2.05 * step and float((err > 0.5 * step).float().mean()) < 5e-3, (float(err.max()), step)


def test_quantised_modules_fuse_the_projections_into_one_gemm_with_quantiser_epilogue(oa, monkeypatch):
    """SURVEY 8f-1 / VERDICT r3 next #5: on the INT8-storage path the three QuantLinear projections (quantized_opt.py:67-75,
    quantized_bert.py:236-238) run as ONE kernel - pair GEMM, weight scale, bias, the three 8-bit output quantisers
    (`ops.proj_quant_i8`); same module outputs and (k, v) cache as with the library GEMM + three quantiser passes up to rare single
    steps (a value on a rounding boundary to within the fp32 accumulation error)."""
    from outeffhop_amd import ops, quantization as Q

    torch.manual_seed(23)
    dev = torch.device("cuda:0")
    B, T, E, H = 2, 128, 768, 12
    org = oa.OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
    qm = oa.QuantizedOPTAttentionWithExtras(org, **_qparams(oa)).to(dev).eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    mask = _decoder_mask(B, T, [T] * B, torch.float32, dev)
    with torch.no_grad():
        for _ in range(3):
            qm(torch.randn(B, T, E, device=dev), attention_mask=mask)
        qm.fix_ranges()
        x = torch.randn(B, T, E, device=dev)
        monkeypatch.setattr(Q, "I8_PLAN", False)  # (this test counts the ops calls of EVERY forward: the prebuilt plan of a repeated forward bypasses them)
        calls = []
        real = ops.proj_quant_i8
        monkeypatch.setattr(ops, "proj_quant_i8", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        out1, _, past1 = qm(x, attention_mask=mask)
        assert len(calls) == 1 and qm.__dict__.get("_fused_proj_calls", 0) == 1 and qm.__dict__.get("_i8_calls", 0) >= 1
        # activations at an odd storage offset (a view 12 bytes into its buffer): the path takes an aligned copy, same outputs
        big = torch.empty(x.numel() + 3, device=dev)
        xo = big[3:].view_as(x).copy_(x)
        assert xo.data_ptr() % 16 != 0
        outo, _, _ = qm(xo, attention_mask=mask)
        assert len(calls) == 2 and torch.equal(outo, out1)
        calls.pop()
        monkeypatch.setattr(Q, "FUSED_PROJ", False)
        out0, _, past0 = qm(x, attention_mask=mask)
        assert len(calls) == 1
    step = float(qm.out_proj.activation_quantizer.quantizer.delta)
    err = (out1 - out0).abs()
    print(f"quantised OPT, fused projections vs library GEMM + quantiser passes: max {float(err.max()) / step:.2f} steps, {float((err > 0.5 * step).float().mean()):.2e} of the outputs apart")
    assert float(err.max()) <= 2.05 * step and float((err > 0.5 * step).float().mean()) < 2e-3
    for n_, lin in ((0, qm.k_proj), (1, qm.v_proj)):
        dp = (past1[n_] - past0[n_]).abs()
        assert float(dp.max()) <= 1.001 * float(lin.activation_quantizer.quantizer.delta) and float((dp > 0).float().mean()) < 1e-4


def test_quantised_bert_uses_the_int8_storage_core(oa, monkeypatch):
    """VERDICT r2 missing #2 / next #5: QuantizedBertSelfAttentionWithExtras (quantized_bert.py:236-238: query / key / value are
    QuantLinear; :363 scores / sqrt(d) quantised before the mask; :374 probabilities; :434 context quantised AFTER the gate and
    the head merge) on the integer matrix cores: with and without a key-padding mask, with a per-token gate; same module output
    as the fake-quant kernels on float values up to rare single steps of the context grid; a fully padded sample gives zeros
    (softmax_1) on both paths."""
    from outeffhop_amd import ops, quantization as Q

    torch.manual_seed(22)
    dev = torch.device("cuda:0")
    B, T = 4, 80

    class C12:
        hidden_size, num_attention_heads, attention_probs_dropout_prob, position_embedding_type, is_decoder = 256, 4, 0.0, "absolute", False
        max_position_embeddings = 512

    for gate in ("nogate", "tok_linear"):
        org = oa.BertSelfAttentionWithExtras(C12(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"], **gate_kwargs(gate)).to(dev).eval()
        qm = oa.QuantizedBertSelfAttentionWithExtras(org, **_qparams(oa)).to(dev).eval()
        qm.set_quant_state(weight_quant=True, act_quant=True)
        fmin = torch.finfo(torch.float32).min
        mask = torch.zeros(B, 1, 1, T, device=dev)
        mask[1, ..., 61:] = fmin
        mask[2, ..., 7:] = fmin
        mask[3] = fmin  # a sample without a single visible key
        with torch.no_grad():
            for _ in range(3):
                qm(torch.randn(B, T, 256, device=dev), attention_mask=mask)
            qm.fix_ranges()
            x = torch.randn(B, T, 256, device=dev)
            monkeypatch.setattr(Q, "I8_PLAN", False)  # (this test counts the ops calls of EVERY forward: the prebuilt plan of a repeated forward bypasses them)
            calls = []
            real = ops.attn_fwd_i8
            monkeypatch.setattr(ops, "attn_fwd_i8", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
            monkeypatch.setattr(Q, "INT8_STORAGE", True)
            o8 = qm(x, attention_mask=mask)[0]
            o8n = qm(x)[0]
            assert len(calls) == 2, "the BERT module did not reach the integer matrix cores"
            monkeypatch.setattr(Q, "INT8_STORAGE", False)
            of = qm(x, attention_mask=mask)[0]
            ofn = qm(x)[0]
            assert len(calls) == 2
            monkeypatch.setattr(ops, "attn_fwd_i8", real)
        step = float(qm.context_act_quantizer.activation_quantizer.quantizer.delta)
        for a, b_ in ((o8, of), (o8n, ofn)):
            err = (a - b_).abs()
            off = float((err > 0.5 * step).float().mean())
            print(f"quantised BERT [{gate}]: int8-storage core vs fake-quant on floats: max {float(err.max()) / step:.2f} steps, {off:.2e} of the outputs apart")
            assert float(err.max()) <= 1.05 * step and off < 2e-3, (gate, float(err.max()), step, off)
        zero_idx = float(qm.context_act_quantizer.activation_quantizer.quantizer.zero_float)
        assert float(o8[3].abs().max()) <= step * abs(round(zero_idx) - zero_idx) + 1e-6  # softmax_1 of a fully padded row: 0 (on the grid)


def test_quantlinear_operand_pair_gemm(oa, monkeypatch):
    """QuantLinear on an fp32 model (the reference's validate precision): the quantised weight is scale * (8-bit integer),
    exact in fp16, and the fp32 input is carried as fp16 operand pairs (oeh_split_pairs), so ONE fp16 GEMM with fp32
    accumulation gives the fp32 linear - at least as close to the exact (float64) result as torch's fp32 GEMM - and the
    module's quantised outputs agree with the fp32-GEMM path except for rare single steps of the output grid."""
    from outeffhop_amd import ops, quantization as Q

    torch.manual_seed(31)
    dev = torch.device("cuda:0")
    lin = torch.nn.Linear(768, 768).to(dev)
    ql = Q.quantize_model(lin, **_qparams(oa)).to(dev).eval()
    ql.quantized_weights()
    x = torch.randn(4, 512, 768, device=dev) * 1.7
    with torch.no_grad():
        ql(x)  # initialises the weight range
        assert ql.pair_gemm_ok(x)
        pairs = ops.split_pairs(x.reshape(-1, 768))
        rec = pairs[:, :768].float() + pairs[:, 768:].float() / 2048.0
        x2 = x.reshape(-1, 768)  # hi + lo 2^-11 reproduces x to 2^-22 relative (values below the fp16 normal range: to 2^-34 absolute)
        assert bool(((rec - x2).abs() <= 2.0 ** -21 * x2.abs() + 2.0 ** -34).all())
        big = torch.tensor([[65520.0, -3.0e5, 1.0, 0.0, 2.0, 3.0, 4.0, 5.0]], device=dev)  # beyond the fp16 range: saturates, never inf
        pb = ops.split_pairs(big).float()
        assert bool(torch.isfinite(pb).all()) and float(pb[0, 0] + pb[0, 8] / 2048.0) == 65520.0 and float(pb[0, 1]) == -65504.0
        got = ql.linear_pairs(x)
        wq, b = ql.get_params()
        exact = torch.nn.functional.linear(x.double(), wq.double(), b.double())
        e_pair = float((got.double() - exact).abs().max())
        e_lib = float((torch.nn.functional.linear(x, wq, b).double() - exact).abs().max())
        assert e_pair <= max(2.0 * e_lib, 1e-5), (e_pair, e_lib)
        # with the output quantiser on (fixed range): same indices as the fp32-GEMM path up to rare boundary cases
        ql.quantized_acts()
        ql.activation_quantizer.set_quant_range(-6.0, 6.0)
        ql.activation_quantizer.fix_ranges()
        a = ql(x)
        monkeypatch.setattr(Q, "PAIR_GEMM", False)
        assert not ql.pair_gemm_ok(x)
        b_ = ql(x)
    step = 12.0 / 255.0
    d = (a - b_).abs()
    assert float(d.max()) <= 1.01 * step and float((d > 0).float().mean()) <= 1e-3


def test_fp32_gated_bert_module_uses_the_in_kernel_predictor(oa, monkeypatch):
    """An fp32 BertSelfAttentionWithExtras with the conditional per-token gate (the reference's validate precision): the predictor
    runs inside the attention kernel (no oeh_gate_fwd launch); output and gate bookkeeping as with the separate gate kernel."""
    from outeffhop_amd import ops

    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    for gate in ("tok_linear", "tok_mlp"):
        m = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"], **gate_kwargs(gate)).to(dev).eval()
        x = torch.randn(4, 96, Cfg.hidden_size, device=dev)
        mask = torch.zeros(4, 1, 1, 96, device=dev)
        mask[1, ..., 70:] = torch.finfo(torch.float32).min
        calls = []
        real = ops.gate_fwd
        monkeypatch.setattr(ops, "gate_fwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with torch.no_grad():
            out = m(x, attention_mask=mask)[0]
            probs = m.last_gate_all_probs.clone()
            assert len(calls) == 0, "the fp32 module launched the separate gate kernel"
            monkeypatch.setattr(ops, "fused_gate_ok", lambda *a, **k: False)
            ref = m(x, attention_mask=mask)[0]
            assert len(calls) == 1
            monkeypatch.undo()
        assert float((out - ref).abs().max()) < 1.5e-3 and float((probs - m.last_gate_all_probs).abs().max()) < 2e-6


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_fused_calibration_vs_observable_16bit(oa, dt):
    """in estimate_ranges state `_calibrate_fused` feeds the estimators fp32 values recomputed from the stored 16-bit
    q / k (what the fused eval kernels quantise), the observable path the score / probability tensors ROUNDED to the storage dtype
    (what the reference's own 16-bit module would hand `np.percentile`).  The two calibrations may differ by one storage ulp of a
    percentile value (the context: of its 16-bit inputs' roundings) - documented in `_calibrate_fused`; bounded here at 2^-10 (fp16) / 2^-6 (bf16) relative on every delta -,
    a forward hook on a quantiser module and a running average without momentum send the module down the observable path."""
    from outeffhop_amd import quantization as Q

    torch.manual_seed(11)
    xs = [torch.randn(2, 96, 128, device="cuda", dtype=dt) for _ in range(3)]
    deltas = {}
    for fused in (True, False):
        keep = Q.FUSED_CALIBRATION
        Q.FUSED_CALIBRATION = fused
        try:
            org = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"])
            torch.manual_seed(12)
            for p in org.parameters():
                torch.nn.init.normal_(p, std=0.08)
            qm = oa.QuantizedOPTAttentionWithExtras(org.cuda().to(dt), **_qparams(oa)).cuda().eval()
            qm.set_quant_state(weight_quant=True, act_quant=True)
            with torch.no_grad():
                for x in xs:
                    qm(x)
            assert (qm.__dict__.get("_fused_calib_calls", 0) == len(xs)) == fused
            deltas[fused] = [float(getattr(qm, n).activation_quantizer.quantizer.delta)
                             for n in ("attn_scores_act_quantizer", "attn_probs_act_quantizer", "context_act_quantizer")]
        finally:
            Q.FUSED_CALIBRATION = keep
    lim = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -6   # (measured: fp16 <= 3e-4; bf16 3e-4 / 3e-3 / 9e-3 for scores / probabilities / context)
    for a, b in zip(deltas[True], deltas[False]):
        assert abs(a - b) <= lim * abs(b), (dt, deltas)
    # a hook on a quantiser module, or no momentum: the observable path (the hook fires; nothing is bypassed)
    org = oa.OPTAttentionWithExtras(128, 2, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"])
    qm = oa.QuantizedOPTAttentionWithExtras(org.cuda().to(dt), **_qparams(oa)).cuda().eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    seen = []
    h = qm.attn_scores_act_quantizer.register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
    with torch.no_grad():
        qm(xs[0])
    h.remove()
    assert seen and qm.__dict__.get("_fused_calib_calls", 0) == 0
    qm.attn_probs_act_quantizer.activation_quantizer.range_estimator.momentum = None   # (no running average: every batch assigns)
    with torch.no_grad():
        qm(xs[1])
    assert qm.__dict__.get("_fused_calib_calls", 0) == 0


def test_int8_storage_core_yields_to_forward_hooks(oa):
    """the INT8-storage fast path reads the QuantLinear weights and quantiser grids directly - the forwards of
    query / key / value (q_proj ...), of the consumer and of the three activation quantisers never run.  The reference's
    `attach_act_hooks` registers a forward hook on every named module; with one present the module path runs and the hook fires."""
    g = load_golden("int8_attn.npz")
    calib = [torch.from_numpy(g[f"calib{i}"]).cuda() for i in range(4)]
    evalx = torch.from_numpy(g["eval"]).cuda()
    bmask = torch.from_numpy(g["bert_mask"]).cuda()
    meta = json.loads(str(g["meta_json"]))[0]
    pre = f"bert{meta['tag']}"
    sd = {k[len(pre) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + ".w.")}
    org = oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING[meta["softmax"]], **gate_kwargs(meta["gate"]))
    org.load_state_dict(sd, strict=True)
    qm = oa.QuantizedBertSelfAttentionWithExtras(org.cuda(), **_qparams(oa)).cuda().eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    with torch.no_grad():
        for c in calib:
            qm(c, attention_mask=bmask)
        qm.fix_ranges()
        plain = qm(evalx, attention_mask=bmask)[0]
        fired = []
        h = qm.key.register_forward_hook(lambda m, i, o: fired.append(1))
        hooked = qm(evalx, attention_mask=bmask)[0]
        h.remove()
    assert fired, "the hook on `key` never fired: the fast path bypassed the module forward"
    step = float(qm.context_act_quantizer.activation_quantizer.quantizer.delta)
    assert float((plain - hooked).abs().max()) <= 1.05 * step


def test_training_backpropagates_through_the_torch_op_path(oa):
    """under the reference's training swap-in (run_clm.py:214-233, run_mlm.py:200-219: modules in train() mode,
    autograd recording) the forward falls back to differentiable torch ops on the GPU instead of raising.  A two-layer toy
    y = x + OPT(x), z = BERT(y), loss = sum(w z) against the forward value and the gradients (input and EVERY parameter: projections,
    gate predictors, the unconditional gate's alpha) captured from the reference (tests/golden/train_grads.npz); inference on the same
    modules keeps the HIP kernels."""
    from tests.test_host_cpu import check_train_toy, run_train_toy

    g = load_golden("train_grads.npz")
    for cj in g["cases_json"]:
        case = json.loads(str(cj))
        la, lb, x, z = run_train_toy(g, case, "cuda")
        check_train_toy(g, case, la, lb, x, z, rtol=1e-3, atol=2e-5)  # fp32 rocBLAS / ATen against the reference's CPU fp32
        la.eval(), lb.eval()
        with torch.no_grad():  # the same modules in inference: the fused kernels (fp32 storage: operand pairs, 5e-4)
            y = x.detach() + la(x.detach(), attention_mask=torch.from_numpy(g["opt_mask"]).cuda())[0]
            z2 = lb(y, attention_mask=torch.from_numpy(g["bert_mask"]).cuda())[0]
        _close(z2, g[f"{case['name']}.z"], case["name"] + " inference after training", dict(atol=2e-3, rtol=2e-3))


def test_frozen_int8_layers_replay_a_prebuilt_plan(oa):
    """Round 5 (the eager quantised modules were host-bound).  Once the ranges are frozen, a repeated forward of
    QuantizedOPT / QuantizedBert runs three PREBUILT launches (quantization._I8LayerPlan) instead of the ~80 Python statements of
    `_int8_storage_core`.  Same bits as the full path; and the plan must notice everything that was baked into it: an in-place weight update,
    a re-fitted range, another input geometry, a padding mask of another batch (pointer patched), a forward hook (bypass forbidden), the feature
    switches - each checked against the full path (I8_PLAN off) on the same module."""
    from outeffhop_amd import quantization as Q

    torch.manual_seed(31)
    dev = torch.device("cuda:0")
    B, T, E, H = 3, 128, 256, 4

    def both(mod, *a, **k):
        """(output through the plan machinery, output of the full path, plan runs during the first)"""
        before = mod.__dict__.get("_i8_plan_runs", 0)
        o1 = mod(*a, **k)
        n = mod.__dict__.get("_i8_plan_runs", 0) - before
        Q.I8_PLAN = False
        try:
            o2 = mod(*a, **k)
        finally:
            Q.I8_PLAN = True
        return o1, o2, n

    # ---- OPT (decoder: int8 core -> out_proj on the int8 context, (k, v) cache values)
    org = oa.OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
    qm = oa.QuantizedOPTAttentionWithExtras(org, **_qparams(oa)).to(dev).eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    mask = _decoder_mask(B, T, [T] * B, torch.float32, dev)
    with torch.no_grad():
        for _ in range(3):
            qm(torch.randn(B, T, E, device=dev), attention_mask=mask)
        qm.fix_ranges()
        x, x2 = torch.randn(B, T, E, device=dev), torch.randn(B, T, E, device=dev)
        first = qm(x, attention_mask=mask)                       # full path, builds the plan
        assert qm.__dict__.get("_oeh_i8_plan") is not None and qm.__dict__.get("_i8_plan_runs", 0) == 0
        (o1, _, p1), (o2, _, p2), n = both(qm, x, attention_mask=mask)
        assert n == 1 and torch.equal(o1, o2) and torch.equal(o1, first[0]) and torch.equal(p1[0], p2[0]) and torch.equal(p1[1], p2[1])
        (o1, _, p1), (o2, _, p2), n = both(qm, x2, attention_mask=mask)     # other data, same geometry: the plan again; fresh outputs
        assert n == 1 and torch.equal(o1, o2) and not torch.equal(o1, first[0]) and o1.data_ptr() != first[0].data_ptr()
        assert torch.equal(first[0], qm(x, attention_mask=mask)[0])        # (the earlier result was not overwritten by later forwards)
        # an in-place weight update / a re-fitted range: noticed (version counters, buffer identity), same result as the full path
        qm.q_proj.weight.mul_(1.01)
        (o1, _, _), (o2, _, _), n = both(qm, x, attention_mask=mask)
        assert n == 0 and torch.equal(o1, o2) and not torch.equal(o1, first[0])
        qz = qm.attn_probs_act_quantizer.activation_quantizer.quantizer
        qz._delta.mul_(1.25)
        (o1, _, _), (o2, _, _), n = both(qm, x, attention_mask=mask)
        assert n == 0 and torch.equal(o1, o2)
        assert both(qm, x, attention_mask=mask)[2] == 1                     # (rebuilt: the next forward replays again)
        # `p.data = other`: identity and version counter both stay - the storage address is watched too
        qm.v_proj.weight.data = qm.v_proj.weight.data * 0.98
        (o1, _, _), (o2, _, _), n = both(qm, x, attention_mask=mask)
        assert n == 0 and torch.equal(o1, o2)
        assert both(qm, x, attention_mask=mask)[2] == 1
        # another geometry
        xs = torch.randn(2, 64, E, device=dev)
        (o1, _, _), (o2, _, _), n = both(qm, xs, attention_mask=_decoder_mask(2, 64, [64, 64], torch.float32, dev))
        assert n == 0 and torch.equal(o1, o2)
        # a forward hook on a bypassed module: the module path must run (the hook fires)
        fired = []
        hnd = qm.k_proj.register_forward_hook(lambda m_, i_, o_: fired.append(1))
        qm(xs, attention_mask=_decoder_mask(2, 64, [64, 64], torch.float32, dev))
        hnd.remove()
        assert fired
        # a feature switch
        qm(x, attention_mask=mask)
        Q.INDEX_GEMM = False
        try:
            before = qm.__dict__.get("_i8_plan_runs", 0)
            qm(x, attention_mask=mask)
            assert qm.__dict__.get("_i8_plan_runs", 0) == before
        finally:
            Q.INDEX_GEMM = True

    # ---- autograd: a plan built under no_grad must NOT be replayed once autograd is recording and the weights / the input
    # require grad.  The FULL path runs then (plan-run counter unchanged) and behaves as it always did: an input that requires grad is
    # refused loudly (the HIP ops are forward-only); with only the parameters requiring grad, an eval-mode QuantLinear works on its
    # cached, detached quantised weights exactly as the reference's does (base_quantized_classes.py: get_params), so the output carries
    # no grad_fn there either - and the full path, not a replay of launches built for another autograd state, is what produced it.
    with torch.no_grad():
        qm(x, attention_mask=mask)
        assert both(qm, x, attention_mask=mask)[2] == 1                     # (a live plan for this geometry)
    before = qm.__dict__.get("_i8_plan_runs", 0)
    with torch.enable_grad():                                               # parameters require grad (the default after construction)
        assert any(p_.requires_grad for p_ in qm.parameters())
        Q.I8_PLAN = False
        try:
            want = qm(x, attention_mask=mask)[0]                            # the full path in this autograd state (non-pair GEMM route)
        finally:
            Q.I8_PLAN = True
        out = qm(x, attention_mask=mask)[0]
        assert torch.equal(out, want) and qm.__dict__.get("_i8_plan_runs", 0) == before
        for p_ in qm.parameters():
            p_.requires_grad_(False)
        xg = x.clone().requires_grad_(True)                                 # frozen weights, an input that requires grad
        with pytest.raises(Q.ops._lib.OehError, match="forward-only"):      # (not a silent replay without grad_fn)
            qm(xg, attention_mask=mask)
        assert qm.__dict__.get("_i8_plan_runs", 0) == before
        assert both(qm, x, attention_mask=mask)[2] == 1                     # nothing requires grad: the plan again, although grad mode is on

    # ---- BERT (key padding: the mask's pointer is patched per call)
    bq = oa.QuantizedBertSelfAttentionWithExtras(oa.BertSelfAttentionWithExtras(Cfg(), softmax_fn=oa.SOFTMAX_MAPPING["softmax1"]).to(dev).eval(),
                                                 **_qparams(oa)).to(dev).eval()
    bq.set_quant_state(weight_quant=True, act_quant=True)
    fmin = torch.finfo(torch.float32).min

    def bmask(lens):
        m = torch.zeros(len(lens), 1, 1, T, device=dev)
        for b_, n_ in enumerate(lens):
            m[b_, ..., n_:] = fmin
        return m

    with torch.no_grad():
        for _ in range(3):
            bq(torch.randn(B, T, 128, device=dev), attention_mask=bmask([T, 90, 40]))
        bq.fix_ranges()
        xb = torch.randn(B, T, 128, device=dev)
        m1, m2 = bmask([T, 90, 40]), bmask([17, T, 64])
        bq(xb, attention_mask=m1)
        (o1,), (o2,), n = both(bq, xb, attention_mask=m1)
        assert n == 1 and torch.equal(o1, o2)
        (o1b,), (o2b,), n = both(bq, xb, attention_mask=m2)                 # another batch's padding through the same plan
        assert n == 1 and torch.equal(o1b, o2b) and not torch.equal(o1b, o1)
        (o1c,), (o2c,), n = both(bq, xb)                                    # no mask: another plan
        assert torch.equal(o1c, o2c)
