#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run ONLY in the build container (where /root/reference is mounted, CPU only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is pure PyTorch eager code with no tests and no golden vectors of
its own, so parity is pinned by what this script captures:
inputs and the reference's outputs, as plain numpy arrays.  Nothing from the
reference's source text is stored; only data.  The GPU box never runs this
script (it has no /root/reference) - it only reads the .npz files.

Shims (applied after `import transformers`, this script only):
  * stub modules timm / timm.models.layers{Swish} / ...activations_me{SwishMe}
    (quantization/hijacker.py:5-6 imports them, nothing on the path uses them)
  * transformers.modeling_utils.apply_chunking_to_forward alias
    (quantized_bert.py:28 - moved to pytorch_utils in transformers 5.x)
  * dummy modeling_opt._expand_mask/_make_causal_mask (quantized_opt.py:12-13,
    only used by the decoder shell, not by the attention class)
"""
import importlib.machinery
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")


def _shim():
    import transformers  # noqa: F401
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    import transformers.models.opt.modeling_opt as mo

    if not hasattr(mu, "apply_chunking_to_forward"):
        mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    for name in ("_expand_mask", "_make_causal_mask"):
        if not hasattr(mo, name):
            setattr(mo, name, lambda *a, **k: None)

    def mk(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
        return m

    class Swish(torch.nn.Module):
        pass

    class SwishMe(torch.nn.Module):
        pass

    mk("timm")
    mk("timm.models")
    layers = mk("timm.models.layers")
    layers.Swish = Swish
    acts = mk("timm.models.layers.activations_me")
    acts.SwishMe = SwishMe


def _f32(t):
    return t.detach().cpu().to(torch.float32).numpy().copy()


def _np(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# --------------------------------------------------------------------------------------
def gen_softmax_rows():
    sys.path.insert(0, os.path.join(REF, "OutEffHop"))
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    g = torch.Generator().manual_seed(1001)
    R, S = 24, 96
    x = torch.randn(R, S, generator=g)
    x[8:16] *= 4.0
    x[16:24] *= 10.0
    fmin = torch.finfo(torch.float32).min
    # rows with masked tails (additive finfo.min as HF does), one fully masked row
    xm = x.clone()
    xm[0, 60:] = fmin
    xm[9, 1:] = fmin
    xm[17, :] = fmin
    edge = {
        "known123": torch.tensor([[1.0, 2.0, 3.0]]),
        "known00": torch.tensor([[0.0, 0.0]]),
        "allmasked": torch.full((1, 4), fmin),
        "neg": torch.tensor([[-20.0, -21.0, -30.0]]),
        "big": torch.tensor([[80.0, 85.0, 88.0, 30.0]]),
        "below_exp_range": torch.tensor([[-90.0, -95.0, -100.0]]),
        "near_exp_range": torch.tensor([[-87.0, -88.0, -88.5]]),
        "single": torch.tensor([[0.3]]),
    }
    arrays = {"x": _np(x), "xm": _np(xm)}
    keys, base, gam, eta = [], [], [], []
    for k, fn in SOFTMAX_MAPPING.items():
        if k == "entmax":
            continue
        keys.append(k)
        if k == "vanilla":
            base.append(0), gam.append(0.0), eta.append(1.0)
        elif k == "softmax1":
            base.append(1), gam.append(0.0), eta.append(1.0)
        else:
            base.append(1 if fn.func.__name__ == "clipped_softmax1" else 0)
            gam.append(float(fn.keywords["gamma"]))
            eta.append(float(fn.keywords["eta"]))
        arrays[f"y[{k}]"] = _np(fn(x, dim=-1))
    for k in ("vanilla", "softmax1", "clipped(-.025:1)", "clippedsoftmax1(-.025:1)", "clipped(0:1.03)"):
        fn = SOFTMAX_MAPPING[k]
        arrays[f"ym[{k}]"] = _np(fn(xm, dim=-1))
        for en, ev in edge.items():
            arrays[f"edge_y[{en}][{k}]"] = _np(fn(ev, dim=-1))
    for en, ev in edge.items():
        arrays[f"edge_x[{en}]"] = _np(ev)
    # pure-fp16 eager behaviour of the reference (documented deviation, SURVEY 7 hard part 3)
    x16 = torch.tensor([[-12.0, -13.0], [1.0, 2.0], [-5.0, -6.0]], dtype=torch.float16)
    arrays["fp16_x"] = _np(x16)
    arrays["fp16_y_softmax1"] = _np(SOFTMAX_MAPPING["softmax1"](x16, dim=-1))
    arrays["keys"] = np.array(keys)
    arrays["key_base"] = np.array(base, dtype=np.int32)
    arrays["key_gamma"] = np.array(gam, dtype=np.float64)
    arrays["key_eta"] = np.array(eta, dtype=np.float64)
    arrays["all_keys_in_order"] = np.array(list(SOFTMAX_MAPPING.keys()))
    save("softmax_rows.npz", **arrays)


# --------------------------------------------------------------------------------------
def gen_fakequant():
    from quantization.quantizers.uniform_quantizers import (
        AsymmetricUniformQuantizer,
        SymmetricUniformQuantizer,
    )

    g = torch.Generator().manual_seed(1002)
    x = torch.cat(
        [
            torch.randn(2048, generator=g) * 3.0,
            torch.rand(1024, generator=g),
            torch.randn(512, generator=g) * 30.0,
            torch.tensor([0.0, -0.0, 1.0, -1.0, 0.5, 1e-9, -1e-9, 255.0, -255.0, 1e6, -1e6]),
        ]
    )
    arrays = {"x": _np(x)}
    ranges = [(-3.0, 5.0), (0.0, 1.0), (-0.37, 0.011), (0.2, 7.0), (-80.0, 60.0), (-1e-12, 1e-12)]
    meta = []
    for i, (lo, hi) in enumerate(ranges):
        for nb in (8, 4):
            q = AsymmetricUniformQuantizer(n_bits=nb)
            q.set_quant_range(lo, hi)
            tag = f"asym{i}_b{nb}"
            arrays[f"{tag}_delta"] = _np(q.delta).astype(np.float64)
            arrays[f"{tag}_zero_float"] = _np(q.zero_float).astype(np.float64)
            arrays[f"{tag}_scale"] = _np(q.scale)
            arrays[f"{tag}_zero_point"] = _np(q.zero_point)
            arrays[f"{tag}_idx"] = _np(q.to_integer_forward(x))
            arrays[f"{tag}_xq"] = _np(q(x))
            meta.append(dict(tag=tag, lo=lo, hi=hi, n_bits=nb, kind="asym"))
    # float64 range scalars as produced by np.percentile
    data = (torch.rand(4, 8, 16, 16, generator=g) ** 3).numpy()
    lo64, hi64 = np.percentile(data, (100 - 99.999, 99.999))
    q = AsymmetricUniformQuantizer(n_bits=8)
    q.set_quant_range(torch.tensor(lo64), torch.tensor(hi64))
    xs = torch.from_numpy(data).reshape(-1)
    arrays["pct_x"] = data.reshape(-1)
    arrays["pct_lo"] = np.float64(lo64)
    arrays["pct_hi"] = np.float64(hi64)
    arrays["pct_delta"] = _np(q.delta)
    arrays["pct_zero_float"] = _np(q.zero_float)
    arrays["pct_idx"] = _np(q.to_integer_forward(xs))
    arrays["pct_xq"] = _np(q(xs))
    assert q.delta.dtype == torch.float64
    # symmetric (weights) - for the QuantLinear shell around the path
    w = torch.randn(64, 48, generator=g) * 0.05
    q = SymmetricUniformQuantizer(n_bits=8)
    q.set_quant_range(w.min(), w.max())
    arrays["sym_w"] = _np(w)
    arrays["sym_delta"] = _np(q.delta)
    arrays["sym_signed"] = np.array(bool(q.signed))
    arrays["sym_wq"] = _np(q(w))
    arrays["meta_json"] = np.array(json.dumps(meta))
    save("fakequant.npz", **arrays)


# --------------------------------------------------------------------------------------
def gen_range_estimators():
    from quantization.range_estimators import CurrentMinMaxEstimator, RunningMinMaxEstimator
    from quantization.quantizers.uniform_quantizers import AsymmetricUniformQuantizer

    g = torch.Generator().manual_seed(1003)
    batches = [torch.randn(2, 4, 16, 16, generator=g) * (1.0 + 0.5 * i) for i in range(4)]
    arrays = {f"batch{i}": _np(b) for i, b in enumerate(batches)}
    for tag, kw in (("pct", dict(percentile=99.999)), ("minmax", dict()), ("pct99", dict(percentile=99.0))):
        est = RunningMinMaxEstimator(**kw)
        q = AsymmetricUniformQuantizer(n_bits=8)
        traj = []
        for b in batches:
            lo, hi = est(b)
            q.set_quant_range(lo, hi)
            traj.append([float(lo), float(hi), float(q.delta), float(q.zero_float)])
        arrays[f"running_{tag}_traj"] = np.array(traj, dtype=np.float64)
    est = CurrentMinMaxEstimator()
    lo, hi = est(batches[0])
    arrays["current_minmax"] = np.array([float(lo), float(hi)])
    save("range_estimators.npz", **arrays)


# --------------------------------------------------------------------------------------
class _Cfg:
    hidden_size = 128
    num_attention_heads = 2
    attention_probs_dropout_prob = 0.1
    position_embedding_type = "absolute"
    is_decoder = False
    max_position_embeddings = 64


GATE_CASES = {
    # name: ctor kwargs (gate type given by name)
    "nogate": dict(),
    "uncond_head": dict(attn_gate_type="unconditional_per_head"),
    "tok_linear": dict(attn_gate_type="conditional_per_token", attn_gate_init=0.25),
    "tok_mlp": dict(attn_gate_type="conditional_per_token", attn_gate_mlp=True),
    "tok_mlp2": dict(attn_gate_type="conditional_per_token", attn_gate_mlp2=True),
    "head_linear": dict(attn_gate_type="conditional_per_head", attn_gate_init=0.5),
    "head_mlp": dict(attn_gate_type="conditional_per_head", attn_gate_mlp=True),
    "tok_allfeat": dict(attn_gate_type="conditional_per_token", attn_gate_linear_all_features=True),
    "tok_linear_ft": dict(attn_gate_type="conditional_per_token", attn_gate_init=0.25, fine_tuning=True),
}


def _randomise_gate(mod, g):
    """Default init leaves some gate params trivial (zeros); perturb all of them, seeded."""
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n.startswith("alpha"):
                p.add_(torch.randn(p.shape, generator=g) * 0.3)


def _bert_mask(B, S, lengths, dtype=torch.float32):
    m = torch.zeros(B, 1, 1, S, dtype=dtype)
    for b, L in enumerate(lengths):
        m[b, :, :, L:] = torch.finfo(dtype).min
    return m


def _opt_mask(B, T, lengths, dtype=torch.float32):
    """HF 4.31 OPTDecoder._prepare_decoder_attention_mask semantics: causal(finfo.min) + padding(finfo.min)."""
    fmin = torch.finfo(dtype).min
    causal = torch.full((T, T), fmin, dtype=dtype).triu(1)[None, None].expand(B, 1, T, T).clone()
    pad = torch.zeros(B, 1, T, T, dtype=dtype)
    for b, L in enumerate(lengths):  # left-aligned tokens, right padding
        pad[b, :, :, L:] = fmin
    return causal + pad  # may hit -inf: the module clamps with torch.max(., finfo.min)


def gen_bert_fp():
    from transformers_language.models.bert_attention import AttentionGateType, BertSelfAttentionWithExtras
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    B, S = 2, 32
    g = torch.Generator().manual_seed(1004)
    hidden = torch.randn(B, S, _Cfg.hidden_size, generator=g)
    mask = _bert_mask(B, S, [32, 20])
    arrays = {"hidden": _np(hidden), "mask": _np(mask)}
    torch.manual_seed(2004)
    base = BertSelfAttentionWithExtras(_Cfg())
    base_sd = {k: v.clone() for k, v in base.state_dict().items()}
    for k, v in base_sd.items():
        arrays[f"w.{k}"] = _np(v)
    cases = []
    for sm in ("vanilla", "softmax1", "clipped(-.025:1)", "clippedsoftmax1(-.025:1)", "clippedsoftmax1(-.0001:1)"):
        cases.append((f"sm[{sm}]", sm, "nogate"))
    for gc in GATE_CASES:
        if gc != "nogate":
            cases.append((f"gate[{gc}]", "softmax1", gc))
    names = []
    for name, sm, gc in cases:
        kw = dict(GATE_CASES[gc])
        if "attn_gate_type" in kw:
            kw["attn_gate_type"] = AttentionGateType[kw["attn_gate_type"]]
        torch.manual_seed(3000 + len(names))
        mod = BertSelfAttentionWithExtras(_Cfg(), softmax_fn=SOFTMAX_MAPPING[sm], **kw)
        mod.load_state_dict(base_sd, strict=False)
        _randomise_gate(mod, g)
        mod.eval()
        for k, v in mod.state_dict().items():
            if k.startswith("alpha"):
                arrays[f"{name}.w.{k}"] = _np(v)
        with torch.no_grad():
            ctx, probs = mod(hidden, attention_mask=mask, output_attentions=True)
            ctx_nomask = mod(hidden)[0]
        arrays[f"{name}.ctx"] = _np(ctx)
        arrays[f"{name}.probs"] = _np(probs)
        arrays[f"{name}.ctx_nomask"] = _np(ctx_nomask)
        if mod.last_gate_avg_prob is not None:
            arrays[f"{name}.last_gate_avg_prob"] = _np(mod.last_gate_avg_prob)
        arrays[f"{name}.gate_scaling_factor"] = np.float64(mod.gate_scaling_factor)
        names.append(json.dumps(dict(name=name, softmax=sm, gate=gc)))
    # alpha (ctor arg) -> clipped softmax with gamma=-alpha/max_seq_length (bert_attention.py:89-92)
    mod = BertSelfAttentionWithExtras(_Cfg(), alpha=4.0, max_seq_length=S)
    mod.load_state_dict(base_sd, strict=False)
    mod.eval()
    with torch.no_grad():
        arrays["alpha4.ctx"] = _np(mod(hidden, attention_mask=mask)[0])
    # skip_attn
    mod = BertSelfAttentionWithExtras(_Cfg(), skip_attn=True)
    arrays["skip.ctx"] = _np(mod(hidden)[0])
    # fp16 module (pure fp16 eager math in the reference) - documented deviation case
    mod = BertSelfAttentionWithExtras(_Cfg(), softmax_fn=SOFTMAX_MAPPING["softmax1"])
    mod.load_state_dict(base_sd, strict=False)
    mod = mod.half().eval()
    with torch.no_grad():
        arrays["half.ctx"] = _np(mod(hidden.half(), attention_mask=_bert_mask(B, S, [32, 20], torch.float16))[0])
    arrays["cases_json"] = np.array(names)
    save("bert_attn_fp.npz", **arrays)


def gen_opt_fp():
    from transformers_language.models.bert_attention import AttentionGateType
    from transformers_language.models.opt_attention import OPTAttentionWithExtras
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    B, T, E, H = 2, 32, 128, 2
    g = torch.Generator().manual_seed(1005)
    hidden = torch.randn(B, T, E, generator=g)
    mask = _opt_mask(B, T, [32, 23])
    arrays = {"hidden": _np(hidden), "mask": _np(mask)}
    torch.manual_seed(2005)
    base = OPTAttentionWithExtras(E, H, is_decoder=True)
    base_sd = {k: v.clone() for k, v in base.state_dict().items()}
    for k, v in base_sd.items():
        arrays[f"w.{k}"] = _np(v)
    cases = []
    for sm in ("vanilla", "softmax1", "clippedsoftmax1(-.025:1)", "clipped(-.003:1.003)"):
        cases.append((f"sm[{sm}]", sm, "nogate"))
    for gc in ("uncond_head", "tok_linear", "tok_mlp", "head_linear", "tok_allfeat", "tok_linear_ft"):
        cases.append((f"gate[{gc}]", "softmax1", gc))
    names = []
    for name, sm, gc in cases:
        kw = dict(GATE_CASES[gc])
        if "attn_gate_type" in kw:
            kw["attn_gate_type"] = AttentionGateType[kw["attn_gate_type"]]
        torch.manual_seed(3100 + len(names))
        mod = OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING[sm], **kw)
        mod.load_state_dict(base_sd, strict=False)
        _randomise_gate(mod, g)
        mod.eval()
        for k, v in mod.state_dict().items():
            if k.startswith("alpha"):
                arrays[f"{name}.w.{k}"] = _np(v)
        with torch.no_grad():
            out, w, past = mod(hidden, attention_mask=mask, output_attentions=True)
            out_nomask = mod(hidden)[0]
        arrays[f"{name}.out"] = _np(out)
        arrays[f"{name}.probs"] = _np(w)
        arrays[f"{name}.out_nomask"] = _np(out_nomask)
        arrays[f"{name}.gate_scaling_factor"] = np.float64(mod.gate_scaling_factor)
        names.append(json.dumps(dict(name=name, softmax=sm, gate=gc)))
    arrays["past_k"] = _np(past[0])
    arrays["past_v"] = _np(past[1])
    # ctor alpha path (opt_attention.py:70-77): `attn_softmax is "softmax1"` is an identity test on a
    # literal; record what the reference actually does for alpha=12, attn_softmax="softmax1".
    mod = OPTAttentionWithExtras(E, H, is_decoder=True, alpha=12.0, max_seq_length=T, attn_softmax="softmax1")
    mod.load_state_dict(base_sd, strict=False)
    mod.eval()
    arrays["alpha12.softmax_fn_name"] = np.array(mod.softmax_fn.func.__name__)
    with torch.no_grad():
        arrays["alpha12.out"] = _np(mod(hidden, attention_mask=mask)[0])
    # the fp16 + softmax1 TypeError - record the exception type name
    mod = OPTAttentionWithExtras(E, H, softmax_fn=SOFTMAX_MAPPING["softmax1"]).half().eval()
    try:
        mod(hidden.half(), attention_mask=_opt_mask(B, T, [32, 23], torch.float16))
        err = "none"
    except Exception as e:  # noqa: BLE001
        err = type(e).__name__
    arrays["half_softmax1_error"] = np.array(err)
    arrays["cases_json"] = np.array(names)
    save("opt_attn_fp.npz", **arrays)


# --------------------------------------------------------------------------------------
def _qparams(percentile=99.999):
    from quantization.range_estimators import RangeEstimators  # noqa: F401
    from transformers_language.quant_configs import get_quant_config
    from transformers_language.utils import val_qparams

    cfg = get_quant_config()
    # validate_clm.py:444-454: asymmetric acts, running_minmax, percentile option
    cfg.act_quant.options = dict(percentile=percentile) if percentile else {}
    qp = val_qparams(cfg)
    qp["quant_dict"] = {}
    return qp


def _capture_quant_io(qmod):
    """Hook the three activation quantizers: record pre-quant input, output, integer indices."""
    rec = {}

    def mk(tag, m):
        def hook(mod, inp, out):
            x = inp[0].detach()
            rec[f"{tag}.in"] = _np(x)
            rec[f"{tag}.out"] = _np(out.detach())
            rec[f"{tag}.idx"] = _np(mod.activation_quantizer.quantizer.to_integer_forward(x)).astype(np.uint8)

        return m.register_forward_hook(hook)

    hs = [
        mk("scores", qmod.attn_scores_act_quantizer),
        mk("probs", qmod.attn_probs_act_quantizer),
        mk("ctx", qmod.context_act_quantizer),
    ]
    return rec, hs


def _dump_quantizers(prefix, qmod, arrays):
    for n, m in qmod.named_modules():
        q = getattr(m, "quantizer", None)
        if n.endswith("range_estimator"):
            continue  # holds a reference to the same quantizer object
        if q is not None and getattr(q, "_delta", None) is not None:
            arrays[f"{prefix}.q.{n}.delta"] = np.float64(float(q._delta))
            zf = getattr(q, "_zero_float", None)
            if zf is not None:
                arrays[f"{prefix}.q.{n}.zero_float"] = np.float64(float(zf))
            sg = getattr(q, "_signed", None)
            if sg is not None:
                arrays[f"{prefix}.q.{n}.signed"] = np.array(bool(sg))


def gen_int8():
    from transformers_language.models.bert_attention import AttentionGateType, BertSelfAttentionWithExtras
    from transformers_language.models.opt_attention import OPTAttentionWithExtras
    from transformers_language.models.quantized_bert import QuantizedBertSelfAttentionWithExtras
    from transformers_language.models.quantized_opt import QuantizedOPTAttentionWithExtras
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    B, S, E, H = 2, 32, 128, 2
    arrays = {}
    meta = []
    g = torch.Generator().manual_seed(1006)
    calib = [torch.randn(B, S, E, generator=g) * (1.0 + 0.1 * i) for i in range(4)]
    evalx = torch.randn(B, S, E, generator=g)
    for i, c in enumerate(calib):
        arrays[f"calib{i}"] = _np(c)
    arrays["eval"] = _np(evalx)
    bmask = _bert_mask(B, S, [32, 20])
    omask = _opt_mask(B, S, [32, 23])
    arrays["bert_mask"] = _np(bmask)
    arrays["opt_mask"] = _np(omask)

    def run(prefix, qmod, fwd):
        qmod.set_quant_state(weight_quant=True, act_quant=True)
        qmod.eval()
        with torch.no_grad():
            for c in calib:
                fwd(qmod, c)
            qmod.fix_ranges()
            _dump_quantizers(prefix, qmod, arrays)
            rec, hs = _capture_quant_io(qmod)
            out = fwd(qmod, evalx)
            for h in hs:
                h.remove()
        for k, v in rec.items():
            arrays[f"{prefix}.{k}"] = v
        arrays[f"{prefix}.out"] = _np(out[0])
        # Q/K/V as the attention core sees them (outputs of the QuantLinear shells)
        with torch.no_grad():
            if hasattr(qmod, "q_proj"):
                arrays[f"{prefix}.q_lin"] = _np(qmod.q_proj(evalx))
                arrays[f"{prefix}.k_lin"] = _np(qmod.k_proj(evalx))
                arrays[f"{prefix}.v_lin"] = _np(qmod.v_proj(evalx))
            else:
                arrays[f"{prefix}.q_lin"] = _np(qmod.query(evalx))
                arrays[f"{prefix}.k_lin"] = _np(qmod.key(evalx))
                arrays[f"{prefix}.v_lin"] = _np(qmod.value(evalx))

    for sm, gc in (("softmax1", "nogate"), ("softmax1", "tok_linear"), ("clippedsoftmax1(-.025:1)", "nogate"), ("vanilla", "nogate")):
        kw = dict(GATE_CASES[gc])
        if "attn_gate_type" in kw:
            kw["attn_gate_type"] = AttentionGateType[kw["attn_gate_type"]]
        tag = f"[{sm}|{gc}]"
        torch.manual_seed(2006)
        org = BertSelfAttentionWithExtras(_Cfg(), softmax_fn=SOFTMAX_MAPPING[sm], **kw)
        _randomise_gate(org, torch.Generator().manual_seed(7))
        for k, v in org.state_dict().items():
            arrays[f"bert{tag}.w.{k}"] = _np(v)
        qmod = QuantizedBertSelfAttentionWithExtras(org, **_qparams())
        run(f"bert{tag}", qmod, lambda m, x: m(x, attention_mask=bmask))
        torch.manual_seed(2007)
        org = OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING[sm], **kw)
        _randomise_gate(org, torch.Generator().manual_seed(8))
        for k, v in org.state_dict().items():
            arrays[f"opt{tag}.w.{k}"] = _np(v)
        qmod = QuantizedOPTAttentionWithExtras(org, **_qparams())
        run(f"opt{tag}", qmod, lambda m, x: m(x, attention_mask=omask))
        meta.append(dict(tag=tag, softmax=sm, gate=gc))
    arrays["meta_json"] = np.array(json.dumps(meta))
    save("int8_attn.npz", **arrays)


# --------------------------------------------------------------------------------------
def gen_stanhop():
    sys.path.insert(0, os.path.join(REF, "STanHop_time_seeries"))
    # cross_models/ and OutEffHop both have a package-less `cross_models`; import fresh
    from cross_models.hopfield import Association, Hopfield, HopfieldPooling

    g = torch.Generator().manual_seed(1007)
    B, L, S, H, E = 3, 7, 5, 4, 16
    q = torch.randn(B, L, H, E, generator=g)
    k = torch.randn(B, S, H, E, generator=g)
    v = torch.randn(B, S, H, E, generator=g)
    arrays = {"q": _np(q), "k": _np(k), "v": _np(v)}
    for mode in ("softmax1", "softmax", "clip"):
        m = Association(mode=mode).eval()
        with torch.no_grad():
            arrays[f"assoc[{mode}]"] = _np(m(q, k, v))
    m = Association(mode="softmax1", scale=0.37).eval()
    with torch.no_grad():
        arrays["assoc[softmax1,scale=0.37]"] = _np(m(q, k, v))
    try:
        Association(mode="clip_softmax1")
        err = "none"
    except Exception as e:  # noqa: BLE001
        err = type(e).__name__
    arrays["clip_softmax1_ctor_error"] = np.array(err)
    # Hopfield / HopfieldPooling modules (V is computed from the projected K: hopfield.py:78)
    torch.manual_seed(2008)
    hop = Hopfield(64, 4, mode="softmax1").eval()
    x = torch.randn(6, 28, 64, generator=g)
    y = torch.randn(6, 11, 64, generator=g)
    for kk, vv in hop.state_dict().items():
        arrays[f"hop.w.{kk}"] = _np(vv)
    with torch.no_grad():
        arrays["hop.x"] = _np(x)
        arrays["hop.y"] = _np(y)
        arrays["hop.self"] = _np(hop(x, x, x))
        arrays["hop.cross"] = _np(hop(x, y, y))
    torch.manual_seed(2009)
    pool = HopfieldPooling(64, 4, num_pattern=3, mode="softmax1").eval()
    with torch.no_grad():
        torch.nn.init.normal_(pool.key, generator=g)
        for kk, vv in pool.state_dict().items():
            arrays[f"pool.w.{kk}"] = _np(vv)
        arrays["pool.out"] = _np(pool(x))
    save("stanhop_assoc.npz", **arrays)
    sys.path.remove(os.path.join(REF, "STanHop_time_seeries"))
    for mname in [m for m in sys.modules if m.startswith("cross_models")]:
        del sys.modules[mname]


def gen_theory_cfg1():
    """BASELINE config 1: theory_verification single Hopfield layer, B=4 S=64 d=32, CPU eager."""
    tv = os.path.join(REF, "theory_verification")
    sys.path.insert(0, tv)
    saved = {m: sys.modules.pop(m) for m in list(sys.modules) if m == "utils" or m.startswith("utils.") or m == "layers"}
    import layers as tv_layers
    from functions import softmax_1 as tv_softmax_1

    torch.manual_seed(0)
    hop = tv_layers.Hopfield(d_model=32, n_heads=1, dropout=0.0).eval()
    R = torch.randn(4, 64, 32)
    arrays = {"R": _np(R)}
    for kk, vv in hop.state_dict().items():
        arrays[f"w.{kk}"] = _np(vv)
    with torch.no_grad():
        arrays["out_softmax"] = _np(hop(R, R))  # reference as-is: Association hard-codes torch.softmax (layers.py:120)
        # config 1 proper: same structure with the reference's softmax_1 substituted for torch.softmax
        q = hop.query_projection(R).view(4, 64, 1, -1)
        kproj = hop.key_projection(R)
        v = hop.value_projection(kproj).view(4, 64, 1, -1)
        k = kproj.view(4, 64, 1, -1)
        scores = torch.einsum("blhe,bshe->bhls", q, k)
        A = tv_softmax_1((1.0 / np.sqrt(32)) * scores, dim=-1)
        V = torch.einsum("bhls,bshd->blhd", A, v).contiguous()
        arrays["out_softmax1"] = _np(hop.out_projection(V.view(4, 64, -1)))
        arrays["assoc_q"] = _np(q)
        arrays["assoc_k"] = _np(k)
        arrays["assoc_v"] = _np(v)
        arrays["assoc_out_softmax1"] = _np(V)
    save("theory_hopfield_cfg1.npz", **arrays)
    sys.path.remove(tv)
    for m in ("layers", "functions"):
        sys.modules.pop(m, None)
    sys.modules.update(saved)


def gen_vit():
    from transformers_language.models.softmax import SOFTMAX_MAPPING
    # vit_attention.py:40-44 defines its OWN AttentionGateType class; the bert one never compares equal to it
    from transformers_language.models.vit_attention import AttentionGateType, ViTSelfAttentionWithExtras

    g = torch.Generator().manual_seed(1010)
    B, N, C, H = 2, 19, 128, 2
    x = torch.randn(B, N, C, generator=g)
    arrays = {"x": _np(x)}
    names = []
    for name, sm, gc in (("sm[softmax1]", "softmax1", "nogate"), ("sm[vanilla]", "vanilla", "nogate"),
                         ("sm[clippedsoftmax1(-.025:1)]", "clippedsoftmax1(-.025:1)", "nogate"),
                         ("gate[tok_linear]", "softmax1", "tok_linear"), ("gate[uncond_head]", "softmax1", "uncond_head")):
        kw = dict(GATE_CASES[gc])
        if "attn_gate_type" in kw:
            kw["attn_gate_type"] = AttentionGateType[kw["attn_gate_type"]]
        torch.manual_seed(2010)
        mod = ViTSelfAttentionWithExtras(C, num_heads=H, qkv_bias=True, softmax_fn=SOFTMAX_MAPPING[sm], **kw)
        _randomise_gate(mod, torch.Generator().manual_seed(9))
        mod.eval()
        for kk, vv in mod.state_dict().items():
            arrays[f"{name}.w.{kk}"] = _np(vv)
        with torch.no_grad():
            arrays[f"{name}.out"] = _np(mod(x))
        arrays[f"{name}.fused_attn_flag"] = np.array(bool(mod.fused_attn))
        names.append(json.dumps(dict(name=name, softmax=sm, gate=gc)))
    arrays["cases_json"] = np.array(names)
    save("vit_attn_fp.npz", **arrays)


def gen_core_cases():
    """Core (B,H,S,d) attention at small sizes straight from reference primitives, fp32 math on
    fp16-rounded inputs - the definition of the fp16 oracle - BERT and OPT op orders."""
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    g = torch.Generator().manual_seed(1011)
    arrays = {}
    B, H, S, d = 2, 3, 80, 64
    q = torch.randn(B, H, S, d, generator=g).half().float()
    k = torch.randn(B, H, S, d, generator=g).half().float()
    v = torch.randn(B, H, S, d, generator=g).half().float()
    arrays.update(q=_np(q), k=_np(k), v=_np(v))
    pad = _bert_mask(B, S, [80, 51])
    arrays["pad_mask"] = _np(pad)
    for sm in ("softmax1", "vanilla", "clippedsoftmax1(-.025:1)"):
        fn = SOFTMAX_MAPPING[sm]
        # BERT order (bert_attention.py:222,265,272,276,292)
        sc = torch.matmul(q, k.transpose(-1, -2)) / np.sqrt(d)
        p = fn(sc + pad, dim=-1)
        arrays[f"bert[{sm}].probs"] = _np(p)
        arrays[f"bert[{sm}].ctx"] = _np(torch.matmul(p, v))
        # OPT order (opt_attention.py:167,204,220-224,232,263): q pre-scaled and (fp16 storage) re-rounded
        qs = (q * d ** -0.5).half().float()
        sc = torch.bmm(qs.view(B * H, S, d), k.view(B * H, S, d).transpose(1, 2)).view(B, H, S, S)
        m = _opt_mask(B, S, [80, 51])
        sc = torch.max(sc + m, torch.tensor(torch.finfo(torch.float32).min))
        p = fn(sc.view(B * H, S, S), dim=-1)
        arrays[f"opt[{sm}].probs"] = _np(p.view(B, H, S, S))
        arrays[f"opt[{sm}].ctx"] = _np(torch.bmm(p, v.view(B * H, S, d)).view(B, H, S, d))
    arrays["opt_mask"] = _np(_opt_mask(B, S, [80, 51]))
    save("core_attn.npz", **arrays)


def gen_sparse_acts():
    """The sort / bisection based activations of the reference's registries (outside the HIP path; torch ops on this side):
    SOFTMAX_MAPPING["entmax"] = entmax15 (vutils/entmax.py), Sparsemax (vutils/sparse_max.py), and STanHop's default
    Association activation EntmaxAlpha (cross_models/entmax.py) incl. the Association output with mode='entmax'/'sparsemax'."""
    from vutils.entmax import entmax15
    from vutils.sparse_max import Sparsemax

    sys.path.insert(0, os.path.join(REF, "STanHop_time_seeries"))
    from cross_models.entmax import EntmaxAlpha, entmax_bisect
    from cross_models.hopfield import Association

    g = torch.Generator().manual_seed(1011)
    x = torch.randn(4, 3, 9, 21, generator=g) * 2.5
    x[0, 0, 0] = 0.0          # a uniform row
    x[0, 0, 1, :3] = 50.0     # a three-way tie far above the rest
    arrays = {"x": _np(x), "entmax15": _np(entmax15(x, dim=-1)), "entmax15_dim1": _np(entmax15(x, dim=1)),
              "sparsemax": _np(Sparsemax(dim=-1)(x)), "bisect_1p3": _np(entmax_bisect(x, 1.3)), "bisect_2p0": _np(entmax_bisect(x, 2.0))}
    # gradients of the bisection form through the reference's own autograd Function: input and alpha
    xg = (torch.randn(2, 3, 5, 11, generator=g) * 1.7).requires_grad_(True)
    ag = EntmaxAlpha()
    with torch.no_grad():
        ag.alpha.fill_(-0.21)
    w = torch.randn(2, 3, 5, 11, generator=g)
    (ag(xg) * w).sum().backward()
    arrays.update(grad_x=_np(xg.detach()), grad_w=_np(w), grad_alpha_param=_np(ag.alpha.detach()), grad_dx=_np(xg.grad), grad_dalpha=_np(ag.alpha.grad))
    ea = EntmaxAlpha().eval()
    with torch.no_grad():
        ea.alpha.fill_(0.37)
        arrays["entmax_alpha_param"] = _np(ea.alpha)
        arrays["entmax_alpha_out"] = _np(ea(x))
        q, k, v = torch.randn(3, 7, 4, 16, generator=g), torch.randn(3, 5, 4, 16, generator=g), torch.randn(3, 5, 4, 16, generator=g)
        arrays.update(q=_np(q), k=_np(k), v=_np(v))
        m = Association(mode="entmax").eval()
        m.softmax.alpha.fill_(0.37)
        arrays["assoc[entmax]"] = _np(m(q, k, v))
        arrays["assoc[sparsemax]"] = _np(Association(mode="sparsemax").eval()(q, k, v))
    save("sparse_acts.npz", **arrays)


def gen_train_grads():
    """Training through the reference's swap-in modules (run_clm.py:214-233, run_mlm.py:200-219): a two-layer toy
    y = x + OPTAttentionWithExtras(x), z = BertSelfAttentionWithExtras(y), loss = sum(w * z), in train() mode with autograd on -
    the forward value and the gradients of the input and of every parameter (attention dropout 0: deterministic)."""
    from transformers_language.models.bert_attention import AttentionGateType, BertSelfAttentionWithExtras
    from transformers_language.models.opt_attention import OPTAttentionWithExtras
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    class Cfg0(_Cfg):
        attention_probs_dropout_prob = 0.0

    B, T, E, H = 2, 32, 128, 2
    g = torch.Generator().manual_seed(1012)
    x0 = torch.randn(B, T, E, generator=g)
    w = torch.randn(B, T, E, generator=g)
    omask = _opt_mask(B, T, [32, 23])
    bmask = _bert_mask(B, T, [32, 20])
    arrays = {"x": _np(x0), "w": _np(w), "opt_mask": _np(omask), "bert_mask": _np(bmask)}
    toys = [("toy0", "softmax1", "tok_linear", "clippedsoftmax1(-.025:1)", "tok_mlp"),
            ("toy1", "vanilla", "uncond_head", "clipped(-.025:1)", "head_linear"),
            ("toy2", "clippedsoftmax1(-.0001:1)", "tok_allfeat", "softmax1", "nogate")]
    names = []
    for i, (name, sm_a, gate_a, sm_b, gate_b) in enumerate(toys):
        def kw_of(gc):
            kw = dict(GATE_CASES[gc])
            if "attn_gate_type" in kw:
                kw["attn_gate_type"] = AttentionGateType[kw["attn_gate_type"]]
            return kw

        torch.manual_seed(4100 + i)
        la = OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING[sm_a], **kw_of(gate_a))
        lb = BertSelfAttentionWithExtras(Cfg0(), softmax_fn=SOFTMAX_MAPPING[sm_b], **kw_of(gate_b))
        _randomise_gate(la, g)
        _randomise_gate(lb, g)
        la.train()
        lb.train()
        x = x0.clone().requires_grad_(True)
        y = x + la(x, attention_mask=omask)[0]
        z = lb(y, attention_mask=bmask)[0]
        (z * w).sum().backward()
        arrays[f"{name}.z"] = _np(z)
        arrays[f"{name}.dx"] = _np(x.grad)
        for tag, mod in (("a", la), ("b", lb)):
            for k, v in mod.state_dict().items():
                arrays[f"{name}.{tag}.w.{k}"] = _np(v)
            for k, p_ in mod.named_parameters():
                arrays[f"{name}.{tag}.g.{k}"] = _np(p_.grad if p_.grad is not None else torch.zeros_like(p_))
                arrays[f"{name}.{tag}.hasgrad.{k}"] = np.array(p_.grad is not None)
        names.append(json.dumps(dict(name=name, softmax_a=sm_a, gate_a=gate_a, softmax_b=sm_b, gate_b=gate_b)))
    arrays["cases_json"] = np.array(names)
    save("train_grads.npz", **arrays)


# --------------------------------------------------------------------------------------
# Round 6: reference outputs at the shapes the FAST kernels run - long rows, 12 heads, the BASELINE cfg4
# calibration - so that those paths are pinned to the reference directly, not only through the oracle.  Inputs and weights come from
# tests/golden/synth.py (seeded numpy streams the tests regenerate); only the reference's outputs are stored.
def _load_synth():
    sys.path.insert(0, OUT)
    import synth

    return synth


def gen_core_long():
    """core_attn_long.npz: (a) OPT order, S = 512, causal, B = 1, H = 2, d = 64 - softmax1 / clippedsoftmax1(-.025:1) / vanilla;
    (b) BERT order, S = 704 keys with left and right key padding, softmax1 / vanilla.  fp32 math on fp16-rounded inputs, as core_attn.npz."""
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    sy = _load_synth()
    arrays = {}
    q, k, v = (torch.from_numpy(a) for a in sy.long_causal_qkv())
    B, H, S, d = q.shape
    mask = torch.from_numpy(sy.opt_decoder_mask(B, S, [S]))
    for sm in ("softmax1", "clippedsoftmax1(-.025:1)", "vanilla"):
        fn = SOFTMAX_MAPPING[sm]
        sc = torch.bmm(q.view(B * H, S, d), k.view(B * H, S, d).transpose(1, 2)).view(B, H, S, S)     # opt_attention.py:204
        sc = torch.max(sc + mask, torch.tensor(torch.finfo(torch.float32).min))                       # :220-224
        p = fn(sc.view(B * H, S, S), dim=-1)                                                          # :232
        arrays[f"opt512[{sm}].ctx"] = _np(torch.bmm(p, v.view(B * H, S, d)).view(B, H, S, d))          # :263
        arrays[f"opt512[{sm}].probs_rowsum"] = _np(p.sum(-1).view(B, H, S))
    q, k, v = (torch.from_numpy(a) for a in sy.long_padded_qkv())
    B, H, S, d = q.shape
    pad = torch.from_numpy(sy.key_padding(B, S, sy.LONG_PAD_LEFT, sy.LONG_PAD_RIGHT)).view(B, 1, 1, S)
    for sm in ("softmax1", "vanilla"):
        fn = SOFTMAX_MAPPING[sm]
        sc = torch.matmul(q, k.transpose(-1, -2)) / np.sqrt(d)                                        # bert_attention.py:222,265
        p = fn(sc + pad, dim=-1)                                                                      # :272,276
        arrays[f"bert704[{sm}].ctx"] = _np(torch.matmul(p, v))                                        # :292
    save("core_attn_long.npz", **arrays)


class _Cfg12:
    hidden_size = 768
    num_attention_heads = 12
    attention_probs_dropout_prob = 0.1
    position_embedding_type = "absolute"
    is_decoder = False
    max_position_embeddings = 512


def _load_synth_weights(mod, sy, seed, **kw):
    shapes = {k_: tuple(v_.shape) for k_, v_ in mod.state_dict().items()}
    sd = {k_: torch.from_numpy(v_) for k_, v_ in sy.state_dict_like(shapes, seed, **kw).items()}
    mod.load_state_dict(sd, strict=True)
    return mod.eval()


def gen_h12():
    """bert_attn_h12.npz / opt_attn_h12.npz: module I/O at E = 768, H = 12, S = 64, B = 2 (BERT-base / OPT-125m widths), one plain and
    one gated case each; weights = synth.state_dict_like over the module's own state_dict names."""
    from transformers_language.models.bert_attention import AttentionGateType, BertSelfAttentionWithExtras
    from transformers_language.models.opt_attention import OPTAttentionWithExtras
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    sy = _load_synth()
    B, S, E, H = sy.H12_B, sy.H12_S, sy.H12_E, sy.H12_H
    # ---- BERT: key padding (32 and 49 visible keys... of 64)
    hidden = torch.from_numpy(sy.h12_hidden(6201))
    mask = torch.from_numpy(sy.key_padding(B, S, [0, 0], [0, 15])).view(B, 1, 1, S)
    arrays, meta = {}, []
    for i, (sm, gc) in enumerate((("softmax1", "nogate"), ("softmax1", "tok_mlp"), ("clipped(-.025:1)", "nogate"))):
        kw = dict(GATE_CASES[gc])
        if "attn_gate_type" in kw:
            kw["attn_gate_type"] = AttentionGateType[kw["attn_gate_type"]]
        mod = _load_synth_weights(BertSelfAttentionWithExtras(_Cfg12(), softmax_fn=SOFTMAX_MAPPING[sm], **kw), sy, 6210 + i, w_std=0.04)
        with torch.no_grad():
            arrays[f"[{sm}|{gc}].ctx"] = _np(mod(hidden, attention_mask=mask)[0])
        if mod.last_gate_avg_prob is not None:
            arrays[f"[{sm}|{gc}].last_gate_avg_prob"] = _np(mod.last_gate_avg_prob)
        meta.append(dict(softmax=sm, gate=gc, seed=6210 + i, w_std=0.04))
    arrays["meta_json"] = np.array(json.dumps(meta))
    save("bert_attn_h12.npz", **arrays)
    # ---- OPT: decoder mask (causal + right padding of sample 1)
    hidden = torch.from_numpy(sy.h12_hidden(6202))
    mask = torch.from_numpy(sy.opt_decoder_mask(B, S, [S, 50]))
    arrays, meta = {}, []
    for i, (sm, gc) in enumerate((("softmax1", "nogate"), ("clippedsoftmax1(-.025:1)", "nogate"), ("softmax1", "tok_linear"))):
        kw = dict(GATE_CASES[gc])
        if "attn_gate_type" in kw:
            kw["attn_gate_type"] = AttentionGateType[kw["attn_gate_type"]]
        mod = _load_synth_weights(OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING[sm], **kw), sy, 6220 + i, w_std=0.04)
        with torch.no_grad():
            out, _, past = mod(hidden, attention_mask=mask)
        arrays[f"[{sm}|{gc}].out"] = _np(out)
        meta.append(dict(softmax=sm, gate=gc, seed=6220 + i, w_std=0.04))
    arrays["meta_json"] = np.array(json.dumps(meta))
    save("opt_attn_h12.npz", **arrays)


def gen_cfg4_calib():
    """cfg4_calib.npz (BASELINE.json config 4): the reference's QuantizedOPTAttentionWithExtras at OPT-125m size
    (E = 768, H = 12, S = 512, B = 16), softmax1, asymmetric 8-bit activations, running_minmax with percentile 99.999 and EMA 0.9
    (validate_clm.py:444-454) over 4 calibration batches (seeds 2000-2003); then for the evaluation batch (seed 2004) the index
    HISTOGRAMS (256 bins) of the three attention quantisers, the output's max-abs / mean-abs and a few sampled output values.
    Scalars and 3 x 256 integers - no tensors.  About two minutes of CPU time and ~6 GB."""
    from transformers_language.models.opt_attention import OPTAttentionWithExtras
    from transformers_language.models.quantized_opt import QuantizedOPTAttentionWithExtras
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    sy = _load_synth()
    B, S, E, H = sy.CFG4_B, sy.CFG4_S, sy.CFG4_E, sy.CFG4_H
    torch.set_num_threads(8)
    org = _load_synth_weights(OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING["softmax1"]), sy, sy.CFG4_WEIGHT_SEED, w_std=0.05)
    qmod = QuantizedOPTAttentionWithExtras(org, **_qparams())
    qmod.set_quant_state(weight_quant=True, act_quant=True)
    qmod.eval()
    mask = torch.from_numpy(sy.opt_decoder_mask(B, S, [S] * B))
    arrays = {}
    with torch.no_grad():
        for i, seed in enumerate(sy.CFG4_CALIB_SEEDS):
            qmod(torch.from_numpy(sy.cfg4_hidden(seed)), attention_mask=mask)
            snap = {}
            _dump_quantizers(f"after{i + 1}", qmod, snap)   # the EMA's trajectory, batch by batch
            arrays.update(snap)
            print(f"  calibration batch {i + 1} done", flush=True)
        qmod.fix_ranges()
        _dump_quantizers("final", qmod, arrays)
        hist = {}

        def mk(tag, m):
            def hook(mod, inp, out):
                idx = mod.activation_quantizer.quantizer.to_integer_forward(inp[0].detach())
                hist[tag] = np.bincount(idx.to(torch.int64).flatten().numpy(), minlength=256).astype(np.int64)
                hist[tag + ".in_absmax"] = np.float64(float(inp[0].abs().max()))

            return m.register_forward_hook(hook)

        hs = [mk("scores", qmod.attn_scores_act_quantizer), mk("probs", qmod.attn_probs_act_quantizer), mk("ctx", qmod.context_act_quantizer)]
        out = qmod(torch.from_numpy(sy.cfg4_hidden(sy.CFG4_EVAL_SEED)), attention_mask=mask)[0]
        for h in hs:
            h.remove()
    for k_, v_ in hist.items():
        arrays[f"eval.hist.{k_}"] = v_
    arrays["eval.out_absmax"] = np.float64(float(out.abs().max()))
    arrays["eval.out_absmean"] = np.float64(float(out.abs().mean()))
    arrays["eval.out_sample"] = _np(out[::2, ::7, ::11])     # 8 x 74 x 70 sampled outputs
    save("cfg4_calib.npz", **arrays)
    torch.set_num_threads(1)


def gen_vit_s16():
    """vit_attn_s16.npz: ViTSelfAttentionWithExtras at ViT-S/16 size (C = 384, 6 heads of 64, N = 197 tokens: the ragged last key tile), B = 2;
    softmax1, clipped softmax1 and a gated case; weights from synth.state_dict_like over the module's own state_dict names."""
    from transformers_language.models.softmax import SOFTMAX_MAPPING
    from transformers_language.models.vit_attention import AttentionGateType, ViTSelfAttentionWithExtras

    sy = _load_synth()
    x = torch.from_numpy(sy.vit_tokens(6301))
    arrays, meta = {}, []
    for i, (sm, gc) in enumerate((("softmax1", "nogate"), ("clippedsoftmax1(-.025:1)", "nogate"), ("softmax1", "tok_linear"))):
        kw = dict(GATE_CASES[gc])
        if "attn_gate_type" in kw:
            kw["attn_gate_type"] = AttentionGateType[kw["attn_gate_type"]]
        mod = _load_synth_weights(ViTSelfAttentionWithExtras(sy.VIT_C, num_heads=sy.VIT_H, qkv_bias=True, softmax_fn=SOFTMAX_MAPPING[sm], **kw), sy, 6310 + i, w_std=0.05)
        with torch.no_grad():
            arrays[f"[{sm}|{gc}].out"] = _np(mod(x))
        meta.append(dict(softmax=sm, gate=gc, seed=6310 + i, w_std=0.05))
    arrays["meta_json"] = np.array(json.dumps(meta))
    save("vit_attn_s16.npz", **arrays)


def gen_bert_int8_calib():
    """bert_int8_calib.npz: the reference's QuantizedBertSelfAttentionWithExtras at BERT-base size (E = 768, H = 12, S = 128, B = 32) with a key-padding mask,
    softmax1: quantiser scalars after each of 4 calibration batches (percentile 99.999, EMA 0.9), then index histograms of the three attention quantisers
    and sampled outputs for the evaluation batch - the BERT twin of cfg4_calib.npz (quantized_bert.py:268-440: scores quantised before the mask, context after
    the gate)."""
    from transformers_language.models.bert_attention import BertSelfAttentionWithExtras
    from transformers_language.models.quantized_bert import QuantizedBertSelfAttentionWithExtras
    from transformers_language.models.softmax import SOFTMAX_MAPPING

    sy = _load_synth()
    B, S = sy.BI8_B, sy.BI8_S
    torch.set_num_threads(8)
    org = _load_synth_weights(BertSelfAttentionWithExtras(_Cfg12(), softmax_fn=SOFTMAX_MAPPING["softmax1"]), sy, sy.BI8_WEIGHT_SEED, w_std=0.05)
    qmod = QuantizedBertSelfAttentionWithExtras(org, **_qparams())
    qmod.set_quant_state(weight_quant=True, act_quant=True)
    qmod.eval()
    lens = sy.bi8_lengths()
    mask = torch.from_numpy(sy.key_padding(B, S, [0] * B, [S - n for n in lens])).view(B, 1, 1, S)
    arrays = {}
    with torch.no_grad():
        for i, seed in enumerate(sy.BI8_CALIB_SEEDS):
            qmod(torch.from_numpy(sy.bi8_hidden(seed)), attention_mask=mask)
            _dump_quantizers(f"after{i + 1}", qmod, arrays)
        qmod.fix_ranges()
        _dump_quantizers("final", qmod, arrays)
        hist = {}

        def mk(tag, m):
            def hook(mod, inp, out):
                idx = mod.activation_quantizer.quantizer.to_integer_forward(inp[0].detach())
                hist[tag] = np.bincount(idx.to(torch.int64).flatten().numpy(), minlength=256).astype(np.int64)

            return m.register_forward_hook(hook)

        hs = [mk("scores", qmod.attn_scores_act_quantizer), mk("probs", qmod.attn_probs_act_quantizer), mk("ctx", qmod.context_act_quantizer)]
        out = qmod(torch.from_numpy(sy.bi8_hidden(sy.BI8_EVAL_SEED)), attention_mask=mask)[0]
        for h in hs:
            h.remove()
    for k_, v_ in hist.items():
        arrays[f"eval.hist.{k_}"] = v_
    arrays["eval.out_absmax"] = np.float64(float(out.abs().max()))
    arrays["eval.out_sample"] = _np(out[::2, ::5, ::13])     # 16 x 26 x 60 sampled outputs
    save("bert_int8_calib.npz", **arrays)
    torch.set_num_threads(1)


def main():
    only = set(sys.argv[1:])
    assert os.path.isdir(REF), "reference not mounted: golden fixtures can only be generated in the build container"
    torch.set_num_threads(1)  # deterministic reduction order for the captured outputs
    _shim()
    gens = [gen_softmax_rows, gen_fakequant, gen_range_estimators, gen_bert_fp, gen_opt_fp, gen_int8, gen_vit, gen_core_cases,
            gen_stanhop, gen_theory_cfg1, gen_sparse_acts, gen_train_grads, gen_core_long, gen_h12, gen_cfg4_calib, gen_vit_s16, gen_bert_int8_calib]
    sys.path.insert(0, os.path.join(REF, "OutEffHop"))
    for fn in gens:  # `make_golden.py gen_vit` regenerates one file
        if not only or fn.__name__ in only:
            fn()


if __name__ == "__main__":
    main()
