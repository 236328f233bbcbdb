"""CPU oracle for the OutEffHop modified-softmax attention hot path (numpy, fp32).

TEST INFRASTRUCTURE ONLY.  This file restates, in plain numpy, the arithmetic of the
reference's hot path so the HIP kernels can be checked against it.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
package (outeffhop_amd/) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
against fixtures under tests/golden/*.npz that were captured by importing the reference
itself (tests/golden/make_golden.py); the reference has no tests or golden vectors of
its own.

Each function cites the reference lines it follows (paths relative to /root/reference).
All arithmetic is IEEE fp32 with the same operation order as the reference's eager ops
(one rounding per torch op, no fused multiply-add), except matmuls whose accumulation
order is BLAS-defined in the reference as well.
"""
from __future__ import annotations

import math
import re
from typing import Dict, Optional, Tuple

import numpy as np

F32 = np.float32
FMIN32 = np.finfo(np.float32).min
FMIN16 = float(np.finfo(np.float16).min)

# ----------------------------------------------------------------------------------------------
# softmax family
# ----------------------------------------------------------------------------------------------


def softmax_vanilla(x: np.ndarray, axis: int = -1) -> np.ndarray:
    """torch.nn.functional.softmax (SOFTMAX_MAPPING["vanilla"],
    OutEffHop/transformers_language/models/softmax.py:23): exp(x-m)/sum."""
    x = np.asarray(x, dtype=F32)
    m = x.max(axis=axis, keepdims=True)
    with np.errstate(invalid="ignore", over="ignore"):
        e = np.exp(x - m, dtype=F32)
        return (e / e.sum(axis=axis, keepdims=True, dtype=F32)).astype(F32)


def softmax_n(x: np.ndarray, n: float = 1.0, axis: int = -1) -> np.ndarray:
    """softmax_n_shifted_zeros (OutEffHop/vutils/softmax_1.py:4-21):
        m = max_j x_j (:11); e = exp(x - m) (:13-15); den = sum e + n*exp(-m) (:16-20); e/den (:21).
    exp(-m) overflows to +inf for m < -88.72 in fp32, which makes the whole row exactly 0 -
    including fully masked rows (m == finfo.min); that behaviour is part of the contract."""
    x = np.asarray(x, dtype=F32)
    m = x.max(axis=axis, keepdims=True)
    with np.errstate(over="ignore", invalid="ignore"):
        e = np.exp(x - m, dtype=F32)
        s = e.sum(axis=axis, keepdims=True, dtype=F32)
        z = np.exp((m * F32(-1.0)).astype(F32), dtype=F32)
        den = (s + z * F32(n)).astype(F32)
        return (e / den).astype(F32)


def softmax_1(x: np.ndarray, axis: int = -1) -> np.ndarray:
    """softmax_1 (OutEffHop/vutils/softmax_1.py:24-28), identical copies at
    STanHop_time_seeries/cross_models/softmax_1.py:27-31, theory_verification/functions.py:36-40."""
    return softmax_n(x, 1.0, axis)


def clip_stretch(p: np.ndarray, gamma: float, eta: float) -> np.ndarray:
    """clip(p*(eta-gamma)+gamma, 0, 1) (softmax.py:12-13,18-19).  (eta-gamma) is formed in Python
    double precision and then multiplies an fp32 tensor, i.e. it is rounded to fp32 once; the
    multiply and the add are two separately rounded fp32 ops."""
    w = F32(eta - gamma)
    g = F32(gamma)
    t = (np.asarray(p, dtype=F32) * w).astype(F32)
    t = (t + g).astype(F32)
    return np.clip(t, F32(0.0), F32(1.0))


def _parse_num(tok: str) -> float:
    return float(tok if not tok.startswith(("-.", ".")) else tok.replace(".", "0.", 1))


def softmax_table() -> Dict[str, Tuple[int, float, float]]:
    """The --attn_softmax registry (softmax.py:22-64) restated as key -> (base, gamma, eta),
    base 0 = vanilla softmax, 1 = softmax_1.  gamma/eta are parsed from the key text except for
    the reference's two literal quirks: "clipped(-.005:1.005)" uses gamma=-0.003 (softmax.py:57)
    and "clippedsoftmax1(-.025:1)" uses eta=1.1 (softmax.py:61).  "entmax" is not on this path."""
    keys = ["vanilla", "softmax1"]
    keys += [f"clipped(0:{e})" for e in ("1.0003", "1.001", "1.002", "1.003", "1.004", "1.01", "1.02", "1.03", "1.1")]
    keys += ["clipped(-.1:1)"]
    keys += [f"clipped({g}:1)" for g in ("-.00001", "-.00003", "-.0001", "-.0003", "-.0005", "-.001", "-.002", "-.0025",
                                         "-.003", "-.004", "-.005", "-.01", "-.015", "-.02", "-.025", "-.03", "-.04")]
    keys += ["clipped(-.001:1.001)", "clipped(-.002:1.002)", "clipped(-.003:1.003)", "clipped(-.005:1.005)",
             "clipped(-.01:1.01)", "clipped(-.03:1.03)", "clipped(-.1:1.1)"]
    keys += ["clippedsoftmax1(-.025:1)", "clippedsoftmax1(-.00001:1)", "clippedsoftmax1(-.0001:1)"]
    table: Dict[str, Tuple[int, float, float]] = {}
    for k in keys:
        if k == "vanilla":
            table[k] = (0, 0.0, 1.0)
        elif k == "softmax1":
            table[k] = (1, 0.0, 1.0)
        else:
            mo = re.fullmatch(r"(clipped|clippedsoftmax1)\(([^:]+):([^)]+)\)", k)
            base = 1 if mo.group(1) == "clippedsoftmax1" else 0
            gamma, eta = _parse_num(mo.group(2)), _parse_num(mo.group(3))
            if k == "clipped(-.005:1.005)":
                gamma = -0.003
            if k == "clippedsoftmax1(-.025:1)":
                eta = 1.1
            table[k] = (base, gamma, eta)
    return table


def apply_softmax(x: np.ndarray, base: int, gamma: float = 0.0, eta: float = 1.0, clip: Optional[bool] = None):
    """SOFTMAX_MAPPING[key](x, dim=-1) for key -> (base, gamma, eta)."""
    p = softmax_n(x, 1.0) if base == 1 else softmax_vanilla(x)
    if clip is None:
        clip = not (gamma == 0.0 and eta == 1.0)
    return clip_stretch(p, gamma, eta) if clip else p


# ----------------------------------------------------------------------------------------------
# per-tensor asymmetric uniform fake-quant
# ----------------------------------------------------------------------------------------------


def fq_grid(delta, zero_float, n_bits: int = 8, eps: float = 1e-8) -> Tuple[np.float32, np.float32, np.float32]:
    """(scale, zero_point, int_max) as fp32 scalars.
    scale = clamp(delta, min=eps) (uniform_quantizers.py:72-74); zero_point = clamp(round(zero_float), 0, 2^n-1)
    (:79-82); delta/zero_float may be float64 0-dim tensors (np.percentile calibration): torch type
    promotion casts 0-dim operands to the fp32 tensor dtype, so rounding them to fp32 first is exact."""
    qmax = F32(2.0 ** n_bits - 1)
    scale = F32(max(float(delta), eps)) if np.asarray(delta).dtype == np.float64 else np.maximum(F32(delta), F32(eps))
    zp = np.clip(np.rint(np.asarray(zero_float, dtype=np.float64)), 0.0, float(qmax))
    return F32(scale), F32(zp), qmax


def fq_index(x: np.ndarray, scale, zp, qmax) -> np.ndarray:
    """to_integer_forward (uniform_quantizers.py:114-115): clamp(round(x/scale) + zp, 0, qmax) with
    torch.round = round-half-to-even (quantizer_utils.py:8-9) and a true IEEE fp32 division."""
    x = np.asarray(x, dtype=F32)
    with np.errstate(over="ignore", invalid="ignore"):
        r = np.rint((x / F32(scale)).astype(F32)).astype(F32)
        return np.clip((r + F32(zp)).astype(F32), F32(0.0), F32(qmax))


def fq_dequant(idx: np.ndarray, scale, zp) -> np.ndarray:
    """x_q = scale * (x_int - zero_point) (uniform_quantizers.py:146)."""
    return (F32(scale) * (np.asarray(idx, dtype=F32) - F32(zp)).astype(F32)).astype(F32)


def fake_quant(x, delta, zero_float, n_bits: int = 8):
    scale, zp, qmax = fq_grid(delta, zero_float, n_bits)
    idx = fq_index(x, scale, zp, qmax)
    return fq_dequant(idx, scale, zp), idx


def quant_range_to_params(x_min, x_max, n_bits: int = 8, eps: float = 1e-8):
    """set_quant_range (uniform_quantizers.py:204-224) + _tensorize_min_max (:173-202):
    x_min = min(x_min, 0); x_max = max(x_max, eps); delta = (x_max-x_min)/(2^n-1);
    zero_float = -x_min/delta.  Python floats become fp32 tensors (:188-189); tensors keep their
    dtype (float64 when they come out of np.percentile, range_estimators.py:91-94)."""
    dt = np.float64 if (isinstance(x_min, np.ndarray) or isinstance(x_min, np.floating)) and np.asarray(x_min).dtype == np.float64 else F32
    lo = np.minimum(dt(x_min), dt(0.0))
    hi = np.maximum(dt(x_max), dt(1.0) * dt(eps))
    qmax = 2.0 ** n_bits - 1  # python float: keeps the tensor dtype
    delta = dt((hi - lo) / dt(qmax))
    zero_float = dt(-lo / delta)
    return delta, zero_float


def sym_weight_quant(w: np.ndarray, n_bits: int = 8, eps: float = 1e-8) -> np.ndarray:
    """SymmetricUniformQuantizer with CurrentMinMax ranges (uniform_quantizers.py:243-310,
    range_estimators.py:71-72) - the weight grid of the QuantLinear shells that feed the path."""
    w = np.asarray(w, dtype=F32)
    lo = np.minimum(w.min(), F32(0.0))
    hi = np.maximum(w.max(), F32(eps))
    signed = bool(lo < 0)
    int_max = F32(2.0 ** (n_bits - (1 if signed else 0)) - 1)
    int_min = F32(-(2.0 ** (n_bits - 1))) if signed else F32(0.0)
    delta = (np.maximum(np.abs(lo), hi) / int_max).astype(F32)
    scale = np.maximum(delta, F32(eps))
    idx = np.clip(np.rint((w / scale).astype(F32)), int_min, int_max).astype(F32)
    return (scale * idx).astype(F32)


class RunningMinMax:
    """RunningMinMaxEstimator (range_estimators.py:77-106), per-tensor branches only:
    percentile -> np.percentile(x, (100-p, p)) over the flattened tensor (:89-94), else min/max (:96-97);
    first call assigns (:99-101), later calls EMA with momentum 0.9 (:103-104)."""

    def __init__(self, momentum: float = 0.9, percentile: Optional[float] = None):
        self.momentum, self.percentile = momentum, percentile
        self.cur_min = self.cur_max = None

    def update(self, x: np.ndarray):
        x = np.asarray(x)
        if self.percentile:
            lo, hi = np.percentile(x, (100 - self.percentile, self.percentile))
        else:
            lo, hi = x.min(), x.max()
        if self.cur_min is None:
            self.cur_min, self.cur_max = lo, hi
        else:
            self.cur_min = (1 - self.momentum) * lo + self.momentum * self.cur_min
            self.cur_max = (1 - self.momentum) * hi + self.momentum * self.cur_max
        return self.cur_min, self.cur_max


# ----------------------------------------------------------------------------------------------
# gate
# ----------------------------------------------------------------------------------------------


def sigmoid(x):
    x = np.asarray(x, dtype=F32)
    with np.errstate(over="ignore"):
        return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32)


def logit(p, eps=1e-16):
    """bert_attention.py:16-18."""
    p = np.clip(p, eps, 1 - eps)
    return -np.log(1 / p - 1)


def gate_values(hidden: np.ndarray, H: int, kind: str, params: dict, per_head_pool: bool = False) -> np.ndarray:
    """Gate probabilities, shape (B,H,T,1) or (B,H,1,1) or (H,1,1)  (bert_attention.py:294-325 =
    opt_attention.py:276-307 = vit_attention.py:226-257).
      kind "unconditional": sigmoid(alpha[H])                                   (:295-297)
      kind "all_features":  sigmoid(Linear(E,H)(x)) permuted to (B,H,T,1)        (:307-311)
      kind "linear"|"mlp":  per head h, fc_h on x[:, :, h*d:(h+1)*d] (the MODULE INPUT split by head,
                            :314-320); conditional_per_head first averages alpha over T (:321-322)
    params: unconditional {"alpha": (H,)}; all_features {"weight": (H,E), "bias": (H,)};
            linear {"w": (H,d), "b": (H,)}; mlp {"w1": (H,m,d), "b1": (H,m), "w2": (H,m), "b2": (H,)}."""
    if kind == "unconditional":
        return sigmoid(params["alpha"]).reshape(-1, 1, 1)
    x = np.asarray(hidden, dtype=F32)
    B, T, E = x.shape
    if kind == "all_features":
        a = (x @ params["weight"].T.astype(F32) + params["bias"].astype(F32)).astype(F32)  # (B,T,H)
        return sigmoid(a).transpose(0, 2, 1)[..., None]
    d = E // H
    xh = x.reshape(B, T, H, d).transpose(0, 2, 1, 3)  # (B,H,T,d)
    if kind == "linear":
        a = np.einsum("bhtd,hd->bht", xh, params["w"].astype(F32)).astype(F32) + params["b"].astype(F32)[None, :, None]
    elif kind == "mlp":
        h1 = np.einsum("bhtd,hmd->bhtm", xh, params["w1"].astype(F32)).astype(F32) + params["b1"].astype(F32)[None, :, None, :]
        h1 = np.maximum(h1, F32(0.0))
        a = np.einsum("bhtm,hm->bht", h1, params["w2"].astype(F32)).astype(F32) + params["b2"].astype(F32)[None, :, None]
    else:
        raise ValueError(kind)
    a = a.astype(F32)[..., None]  # (B,H,T,1)
    if per_head_pool:
        a = a.mean(axis=2, keepdims=True, dtype=F32)
    return sigmoid(a)


# ----------------------------------------------------------------------------------------------
# the attention core  (B,H,S,d) -> (B,H,S,d)
# ----------------------------------------------------------------------------------------------


def causal_additive(Sq: int, Sk: int, min_value: float) -> np.ndarray:
    """(Sq,Sk) additive causal mask, min_value strictly above the (Sk-Sq)-shifted diagonal
    (HF 4.31 _make_causal_mask; call site quantized_opt.py:8-14)."""
    i = np.arange(Sq)[:, None] + (Sk - Sq)
    j = np.arange(Sk)[None, :]
    return np.where(j > i, F32(min_value), F32(0.0)).astype(F32)


def attn_core(
    q: np.ndarray,
    k: np.ndarray,
    v: np.ndarray,
    *,
    scale: float = 1.0,
    scale_is_divisor: bool = False,
    base: int = 1,
    gamma: float = 0.0,
    eta: float = 1.0,
    clip: bool = False,
    pad_mask: Optional[np.ndarray] = None,  # (B,Sk) additive
    full_mask: Optional[np.ndarray] = None,  # (B,1,Sq,Sk) additive
    causal: bool = False,
    mask_min: float = float(FMIN32),
    clamp_min: bool = False,
    fq_scores=None,  # (delta, zero_float[, n_bits])
    fq_probs=None,
    fq_ctx=None,
    gate: Optional[np.ndarray] = None,  # broadcastable to (B,H,Sq,1); already multiplied by gate_scaling_factor
    ctx_quant_before_gate: bool = True,
    want: Tuple[str, ...] = (),
):
    """One attention core in the reference's op order.  q,k,v: (B,H,S,d) arrays (any float dtype; cast to fp32).

    BERT order (bert_attention.py:222-292): scores = q@k^T (:222); scores / sqrt(d) (:265, pass
    scale=sqrt(d), scale_is_divisor=True); [scores fake-quant, quantized_bert.py:363]; + mask (:272);
    softmax_fn (:276); [probs fake-quant, quantized_bert.py:374]; probs@v (:292); gate (:294-327);
    [context fake-quant AFTER gate, quantized_bert.py:434 -> ctx_quant_before_gate=False].
    OPT order (opt_attention.py:167-263): q arrives pre-scaled (:167) so scale=1; bmm (:204);
    [scores fake-quant, quantized_opt.py:154]; + (B,1,T,S) mask then max(., finfo.min) (:220-223 ->
    clamp_min=True); softmax_fn (:232); [probs fake-quant, quantized_opt.py:182]; bmm (:263);
    [context fake-quant BEFORE gate, quantized_opt.py:210]; gate (:276-309).
    ViT (vit_attention.py:71-75 / :215-222) and Association (hopfield.py:47-49) multiply by `scale`.
    Returns ctx (B,H,Sq,d) fp32, plus a dict of requested intermediates ("scores","scores_idx",
    "probs","probs_idx","ctx_idx","ctx_prequant")."""
    q = np.asarray(q, dtype=F32)
    k = np.asarray(k, dtype=F32)
    v = np.asarray(v, dtype=F32)
    B, H, Sq, _ = q.shape
    Sk = k.shape[2]
    out = {}
    s = np.matmul(q, np.swapaxes(k, -1, -2)).astype(F32)
    if scale_is_divisor:
        s = (s / F32(scale)).astype(F32)
    elif scale != 1.0:
        s = (s * F32(scale)).astype(F32)
    if fq_scores is not None:
        s, idx = fake_quant(s, *fq_scores)
        if "scores_idx" in want:
            out["scores_idx"] = idx.astype(np.uint8)
    if "scores" in want:
        out["scores"] = s.copy()
    with np.errstate(over="ignore", invalid="ignore"):
        if pad_mask is not None:
            s = (s + np.asarray(pad_mask, dtype=F32)[:, None, None, :]).astype(F32)
        if full_mask is not None:
            s = (s + np.asarray(full_mask, dtype=F32)).astype(F32)
        if causal:
            s = (s + causal_additive(Sq, Sk, mask_min)[None, None]).astype(F32)
        if clamp_min:
            s = np.maximum(s, F32(mask_min))
    p = apply_softmax(s, base, gamma, eta, clip)
    if fq_probs is not None:
        p, idx = fake_quant(p, *fq_probs)
        if "probs_idx" in want:
            out["probs_idx"] = idx.astype(np.uint8)
    if "probs" in want:
        out["probs"] = p.copy()
    ctx = np.matmul(p, v).astype(F32)
    if fq_ctx is not None and ctx_quant_before_gate:
        if "ctx_prequant" in want:
            out["ctx_prequant"] = ctx.copy()
        ctx, idx = fake_quant(ctx, *fq_ctx)
        if "ctx_idx" in want:
            out["ctx_idx"] = idx.astype(np.uint8)
    if gate is not None:
        ctx = (ctx * np.asarray(gate, dtype=F32)).astype(F32)
    if fq_ctx is not None and not ctx_quant_before_gate:
        if "ctx_prequant" in want:
            out["ctx_prequant"] = ctx.copy()
        ctx, idx = fake_quant(ctx, *fq_ctx)
        if "ctx_idx" in want:
            out["ctx_idx"] = idx.astype(np.uint8)
    return (ctx, out) if want else ctx


# ----------------------------------------------------------------------------------------------
# module-level restatements (projection shells are plain Linear layers)
# ----------------------------------------------------------------------------------------------


def linear(x, w, b=None):
    y = np.matmul(np.asarray(x, dtype=F32), np.asarray(w, dtype=F32).T).astype(F32)
    return y if b is None else (y + np.asarray(b, dtype=F32)).astype(F32)


def split_heads(x, H):  # (B,S,E) -> (B,H,S,d)   bert_attention.py:164-167 / opt_attention.py:146-147
    B, S, E = x.shape
    return x.reshape(B, S, H, E // H).transpose(0, 2, 1, 3)


def merge_heads(x):  # (B,H,S,d) -> (B,S,E)   bert_attention.py:335-337 / opt_attention.py:318-322
    B, H, S, d = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, S, H * d)


def gate_params_from_state(sd: dict, H: int, prefix: str = "alpha"):
    """Map reference parameter names (alpha | alpha.{h}.weight|bias | alpha.{h}.0|2.* | alpha.weight|bias)
    (bert_attention.py:119-162) to (kind, params) for gate_values()."""
    if prefix in sd:
        return "unconditional", {"alpha": sd[prefix]}
    if f"{prefix}.weight" in sd:
        return "all_features", {"weight": sd[f"{prefix}.weight"], "bias": sd[f"{prefix}.bias"]}
    if f"{prefix}.0.weight" in sd:
        return "linear", {
            "w": np.stack([sd[f"{prefix}.{h}.weight"][0] for h in range(H)]),
            "b": np.stack([sd[f"{prefix}.{h}.bias"][0] for h in range(H)]),
        }
    if f"{prefix}.0.0.weight" in sd:
        return "mlp", {
            "w1": np.stack([sd[f"{prefix}.{h}.0.weight"] for h in range(H)]),
            "b1": np.stack([sd[f"{prefix}.{h}.0.bias"] for h in range(H)]),
            "w2": np.stack([sd[f"{prefix}.{h}.2.weight"][0] for h in range(H)]),
            "b2": np.stack([sd[f"{prefix}.{h}.2.bias"][0] for h in range(H)]),
        }
    return None, None


def bert_self_attention(sd, hidden, H, *, mask=None, base=1, gamma=0.0, eta=1.0, clip=False,
                        per_head_pool=False, gate_scaling=1.0, fq=None, want=()):
    """BertSelfAttentionWithExtras.forward (bert_attention.py:169-343) / the quantised variant
    (quantized_bert.py:268-440; fq = dict(scores=..., probs=..., ctx=...), gate without scaling :422)."""
    q = split_heads(linear(hidden, sd["query.weight"], sd["query.bias"]), H)
    k = split_heads(linear(hidden, sd["key.weight"], sd["key.bias"]), H)
    v = split_heads(linear(hidden, sd["value.weight"], sd["value.bias"]), H)
    d = q.shape[-1]
    kind, gp = gate_params_from_state(sd, H)
    g = None
    if kind is not None:
        g = gate_values(hidden, H, kind, gp, per_head_pool)
        g = (g * F32(gate_scaling)).astype(F32) if kind != "unconditional" else g
    fq = fq or {}
    res = attn_core(q, k, v, scale=math.sqrt(d), scale_is_divisor=True, base=base, gamma=gamma, eta=eta, clip=clip,
                    pad_mask=None if mask is None else np.asarray(mask).reshape(mask.shape[0], -1),
                    gate=g, fq_scores=fq.get("scores"), fq_probs=fq.get("probs"), fq_ctx=None, want=want)
    ctx, extra = res if want else (res, {})
    ctx = merge_heads(ctx)
    if fq.get("ctx") is not None:  # after gate AND head merge (quantized_bert.py:430-434)
        if "ctx_prequant" in want:
            extra["ctx_prequant"] = ctx.copy()
        ctx, idx = fake_quant(ctx, *fq["ctx"])
        extra["ctx_idx"] = idx.astype(np.uint8)
    return (ctx, extra) if want else ctx


def opt_attention(sd, hidden, H, *, mask=None, base=1, gamma=0.0, eta=1.0, clip=False, per_head_pool=False,
                  gate_scaling=1.0, fq=None, want=(), qkv_override=None, skip_out_proj=False):
    """OPTAttentionWithExtras.forward (opt_attention.py:149-326) / quantised (quantized_opt.py:94-274)."""
    E = hidden.shape[-1]
    d = E // H
    if qkv_override is None:
        ql = (linear(hidden, sd["q_proj.weight"], sd.get("q_proj.bias")) * F32(d ** -0.5)).astype(F32)  # :167
        kl = linear(hidden, sd["k_proj.weight"], sd.get("k_proj.bias"))
        vl = linear(hidden, sd["v_proj.weight"], sd.get("v_proj.bias"))
    else:
        ql, kl, vl = qkv_override
        ql = (np.asarray(ql, dtype=F32) * F32(d ** -0.5)).astype(F32)
    q, k, v = split_heads(ql, H), split_heads(kl, H), split_heads(vl, H)
    kind, gp = gate_params_from_state(sd, H)
    g = None
    if kind is not None:
        g = gate_values(hidden, H, kind, gp, per_head_pool)
        g = (g * F32(gate_scaling)).astype(F32) if kind != "unconditional" else g
    fq = fq or {}
    res = attn_core(q, k, v, scale=1.0, base=base, gamma=gamma, eta=eta, clip=clip, full_mask=mask,
                    clamp_min=mask is not None, gate=g, fq_scores=fq.get("scores"), fq_probs=fq.get("probs"),
                    fq_ctx=fq.get("ctx"), ctx_quant_before_gate=True, want=want)
    ctx, extra = res if want else (res, {})
    ctx = merge_heads(ctx)
    out = ctx if skip_out_proj else linear(ctx, sd["out_proj.weight"], sd.get("out_proj.bias"))
    return (out, extra) if want else out


def association(q, k, v, *, scale=None, base=1, gamma=0.0, eta=1.0, clip=False):
    """Association.forward (STanHop_time_seeries/cross_models/hopfield.py:42-51;
    theory_verification/layers.py:107-123): layouts (B,L,H,E),(B,S,H,E),(B,S,H,D) -> (B,L,H,D)."""
    E = q.shape[-1]
    sc = scale or 1.0 / math.sqrt(E)
    ctx = attn_core(np.transpose(q, (0, 2, 1, 3)), np.transpose(k, (0, 2, 1, 3)), np.transpose(v, (0, 2, 1, 3)),
                    scale=sc, base=base, gamma=gamma, eta=eta, clip=clip)
    return np.ascontiguousarray(np.transpose(ctx, (0, 2, 1, 3)))
