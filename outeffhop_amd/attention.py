"""Shared host-side pieces of the drop-in attention modules: the gate-type enum, gate construction /
evaluation, and the dispatch between the fused HIP kernel and the unfused (observable) path.

Reference anchors: AttentionGateType / logit  OutEffHop/transformers_language/models/bert_attention.py:16-25
                   gate definitions            bert_attention.py:119-162 (= opt_attention.py:105-144, vit_attention.py:152-195)
                   gate evaluation             bert_attention.py:294-331
"""
from __future__ import annotations

import math
import weakref
from enum import Flag
from typing import Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .ops import AttnFakeQuant, SoftmaxSpec
from .softmax import spec_of


class BaseEnumOptions(Flag):
    """Enum whose str() is the member name and that can list its names (quantization/utils.py:35-41)."""

    def __str__(self):
        return self.name

    @classmethod
    def list_names(cls):
        return [m.name for m in cls]


class AttentionGateType(BaseEnumOptions):
    none = 0
    unconditional_per_head = 1
    conditional_per_head = 2
    conditional_per_token = 3


def logit(p, eps=1e-16):
    p = np.clip(p, eps, 1 - eps)
    return -np.log(1 / p - 1)


_CONDITIONAL = (AttentionGateType.conditional_per_head, AttentionGateType.conditional_per_token)


def build_gate(num_heads: int, head_dim: int, model_dim: int, gate_type, gate_init, use_mlp: bool, use_mlp2: bool,
               all_features: bool, fine_tuning: bool, ft_std: float):
    """Create the `alpha` member exactly as the reference names/initialises it:
    Parameter[H] | Linear(E,H) | ModuleList of per-head Linear(d,1) / Sequential(Linear,ReLU,Linear) | None."""
    if gate_type == AttentionGateType.unconditional_per_head:
        return nn.Parameter(torch.zeros(num_heads), requires_grad=True)
    if gate_type not in _CONDITIONAL:
        return None
    if all_features:
        return nn.Linear(model_dim, num_heads, bias=True)
    heads = []
    for _ in range(num_heads):
        if use_mlp or use_mlp2:
            width = head_dim // 4 if use_mlp else head_dim
            heads.append(nn.Sequential(nn.Linear(head_dim, width, bias=True), nn.ReLU(), nn.Linear(width, 1, bias=True)))
            continue
        fc = nn.Linear(head_dim, 1, bias=True)
        if gate_init is not None:
            nn.init.constant_(fc.bias, float(logit(gate_init)))
        if fine_tuning:
            nn.init.normal_(fc.weight, mean=0.0, std=ft_std)
        heads.append(fc)
    return nn.ModuleList(heads)


class GateBookkeeping:
    """`last_gate_avg_prob` (bert_attention.py:329-331: the per-head mean of the gate probabilities, read by the training scripts'
    logging) as a LAZY attribute: the forward pass records the gate tensor, the two reductions run when somebody reads the
    value - not as two extra launches in every layer of every forward.  Assignment (None, a tensor) works as before."""

    @property
    def last_gate_avg_prob(self):
        src = self.__dict__.get("_oeh_gate_avg_src")
        if src is not None:
            gate, num_heads = src
            self.__dict__["_oeh_gate_avg"] = gate.mean(dim=0).view(num_heads, -1).mean(dim=1)
            self.__dict__["_oeh_gate_avg_src"] = None
        return self.__dict__.get("_oeh_gate_avg")

    @last_gate_avg_prob.setter
    def last_gate_avg_prob(self, value):
        self.__dict__["_oeh_gate_avg"] = value
        self.__dict__["_oeh_gate_avg_src"] = None


def _note_gate(mod: nn.Module, gate: torch.Tensor, num_heads: int) -> None:
    mod.last_gate_all_probs = gate
    if isinstance(mod, GateBookkeeping):
        mod.__dict__["_oeh_gate_avg_src"] = (gate, num_heads)
    else:
        mod.last_gate_avg_prob = gate.mean(dim=0).view(num_heads, -1).mean(dim=1)


_warned_autograd = []


def autograd_needed(mod: nn.Module, *ts) -> bool:
    """True when this forward has to be differentiable: autograd is recording and an input or a parameter of the module requires
    grad (training under the reference's swap-in: run_clm.py:214-233, run_mlm.py:200-219).  The modules then take the observable
    torch-op path (`unfused_core`, `softmax.softmax_autograd`, `gate_autograd`) - slower, on the GPU - because the HIP kernels are
    forward-only; under torch.no_grad() / inference_mode() nothing changes.  Says so once."""
    if not torch.is_grad_enabled():
        return False
    if not (any(t is not None and t.requires_grad for t in ts) or any(p.requires_grad for p in mod.parameters())):
        return False
    if not _warned_autograd:
        import warnings

        _warned_autograd.append(True)
        warnings.warn("outeffhop_amd: autograd is recording - this forward runs the differentiable torch-op path, not the fused HIP "
                      "kernels (forward-only); wrap inference in torch.no_grad()", RuntimeWarning, stacklevel=3)
    return True


def gate_autograd(mod: nn.Module, hidden_states: torch.Tensor, num_heads: int) -> Optional[torch.Tensor]:
    """The gate as the reference's torch ops (bert_attention.py:294-331 = opt_attention.py:265-304, vit_attention.py:241-262), so
    that gradients reach `alpha`: probabilities broadcastable to (B,H,T,1) WITHOUT the scaling factor, bookkeeping attributes set."""
    gt = mod.attn_gate_type
    if gt == AttentionGateType.unconditional_per_head:
        gate = torch.sigmoid(mod.alpha)
        mod.last_gate_avg_prob = gate.view(-1)
        return gate.view(1, -1, 1, 1)
    if gt not in _CONDITIONAL:
        return None
    if mod.attn_gate_linear_all_features:
        gate = torch.sigmoid(mod.alpha(hidden_states)).permute(0, 2, 1).contiguous().unsqueeze(3)
    else:
        x = hidden_states.view(hidden_states.shape[:-1] + (num_heads, -1)).permute(0, 2, 1, 3)  # (B,H,T,d)
        logits = []
        for h in range(num_heads):
            a = mod.alpha[h](x[:, h, ...])  # (B,T,1)
            if gt == AttentionGateType.conditional_per_head:
                a = a.mean(dim=1, keepdim=True)  # (B,1,1)
            logits.append(a)
        gate = torch.sigmoid(torch.stack(logits, dim=1))  # (B,H,*,1)
    mod.last_gate_all_probs = gate
    mod.last_gate_avg_prob = gate.mean(dim=0).view(num_heads, -1).mean(dim=1)
    return gate


class GateState:
    """Evaluates the gate with HIP kernels and keeps the reference's bookkeeping attributes."""

    @staticmethod
    def packed_weights(mod: nn.Module):
        """Per-head predictor weights stacked as (H,d)/(H) or (H,m,d)/(H,m)/(H,m)/(H); cached until a parameter changes."""
        heads = mod.alpha
        sig = tuple((p.data_ptr(), p._version) for p in heads.parameters())
        cache = getattr(mod, "_oeh_gate_cache", None)
        if cache is not None and cache[0] == sig:
            return cache[1]
        with torch.no_grad():
            if isinstance(heads[0], nn.Linear):
                packed = (torch.stack([h.weight[0] for h in heads]).float().contiguous(),
                          torch.stack([h.bias[0] for h in heads]).float().contiguous(), None, None)
            else:
                packed = (torch.stack([h[0].weight for h in heads]).float().contiguous(),
                          torch.stack([h[0].bias for h in heads]).float().contiguous(),
                          torch.stack([h[2].weight[0] for h in heads]).float().contiguous(),
                          torch.stack([h[2].bias[0] for h in heads]).float().contiguous())
        mod._oeh_gate_cache = (sig, packed)
        return packed

    @staticmethod
    def predictor(mod: nn.Module, hidden_states: torch.Tensor, num_heads: int, scaling: float):
        """The conditional per-token gate as an `ops.GatePredictor` for evaluation INSIDE the attention kernel, or None
        when the module's gate is of another kind (then `evaluate`).  Bookkeeping attributes are set from the
        predictor's `out` tensor by `finish_predictor` after the launch."""
        if mod.attn_gate_type != AttentionGateType.conditional_per_token or mod.attn_gate_linear_all_features:
            return None
        if hidden_states.dtype not in (torch.float16, torch.bfloat16, torch.float32) or hidden_states.dim() != 3 or hidden_states.stride(2) != 1:
            return None
        w1, b1, w2, b2 = GateState.packed_weights(mod)
        B, T, _ = hidden_states.shape
        out = torch.empty((B, num_heads, T), dtype=torch.float32, device=hidden_states.device)
        return ops.GatePredictor(hidden_states, w1, b1, w2, b2, scaling=float(scaling), out=out)

    @staticmethod
    def finish_predictor(mod: nn.Module, gp, num_heads: int) -> None:
        _note_gate(mod, gp.out.unsqueeze(3), num_heads)

    @staticmethod
    def evaluate(mod: nn.Module, hidden_states: torch.Tensor, num_heads: int) -> Optional[torch.Tensor]:
        """Gate probabilities, broadcastable to (B,H,T,1), fp32, WITHOUT the scaling factor; sets
        last_gate_avg_prob / last_gate_all_probs like bert_attention.py:299,329-331."""
        gt = mod.attn_gate_type
        if gt != AttentionGateType.none and ops.grad_recording(hidden_states, *(mod.alpha.parameters() if isinstance(mod.alpha, nn.Module) else (mod.alpha,))):
            ops._need_gpu(hidden_states, allow_grad=True)
            return gate_autograd(mod, hidden_states, num_heads)
        if gt == AttentionGateType.unconditional_per_head:
            gate = torch.sigmoid(mod.alpha.float())
            mod.last_gate_avg_prob = gate.view(-1)
            return gate.view(1, -1, 1, 1)
        if gt not in _CONDITIONAL:
            return None
        if mod.attn_gate_linear_all_features:
            gate = torch.sigmoid(mod.alpha(hidden_states).float()).permute(0, 2, 1).contiguous().unsqueeze(3)
        else:
            w1, b1, w2, b2 = GateState.packed_weights(mod)
            gate = ops.gate_fwd(hidden_states, num_heads, w1, b1, w2, b2,
                                per_head_pool=(gt == AttentionGateType.conditional_per_head), scaling=1.0)
        _note_gate(mod, gate, num_heads)
        return gate


# ---- fused q/k/v projection: the three E->E Linears of a self-attention layer as ONE library GEMM with the
# concatenated weight, returning strided (B,S,E) views of its (B,S,3E) output (the attention kernel takes strides, so there is
# no copy either side).  Inference only: parameters keep the reference's names (query/key/value, q_proj/k_proj/v_proj) and the
# concatenation is cached per module and rebuilt when a weight changes (tensor version counters / storage pointers).
FUSE_QKV = True


def fused_qkv(owner: nn.Module, x: torch.Tensor, lq: nn.Linear, lk: nn.Linear, lv: nn.Linear, q_scale: float = 1.0):
    """(q, k, v) = (lq(x) * q_scale, lk(x), lv(x)) from one GEMM, or None when the fused form does not apply (autograd on,
    QuantLinear / hooked / unequal layers, q_scale not a power of two - folding it into the weights is exact only then)."""
    if not FUSE_QKV or torch.is_grad_enabled() or not x.is_cuda:
        return None
    if not (type(lq) is nn.Linear and type(lk) is nn.Linear and type(lv) is nn.Linear) or has_hooks(lq, lk, lv):
        return None
    ws = (lq.weight, lk.weight, lv.weight)
    bs = (lq.bias, lk.bias, lv.bias)
    if not (ws[0].shape == ws[1].shape == ws[2].shape) or ws[0].dtype != x.dtype or any((b is None) != (bs[0] is None) for b in bs):
        return None
    mant, _ = math.frexp(q_scale)
    if mant != 0.5:
        return None
    key = tuple((t._version, t.data_ptr()) for t in ws + tuple(b for b in bs if b is not None)) + (q_scale, x.dtype)
    cache = owner.__dict__.get("_oeh_qkv_cache")
    triple = triple_gemm_ok(x, lq, lk, lv)
    key = key + (triple,)
    if cache is None or cache[0] != key:
        w = torch.cat([ws[0].detach() * q_scale, ws[1].detach(), ws[2].detach()], dim=0).contiguous()
        b = None if bs[0] is None else torch.cat([bs[0].detach() * q_scale, bs[1].detach(), bs[2].detach()], dim=0).contiguous()
        cache = (key, triple_weights(w, b), None) if triple else (key, w, b)
        owner.__dict__["_oeh_qkv_cache"] = cache
    if triple:  # fp32 model: one fp16 GEMM on operand triples (bias inside), fp32-accurate
        y = torch.mm(ops.split_triples(x.reshape(-1, x.shape[-1])), cache[1], out_dtype=torch.float32).view(*x.shape[:-1], cache[1].shape[1])
    else:
        y = torch.nn.functional.linear(x, cache[1], cache[2])
    e = ws[0].shape[0]
    return y[..., :e], y[..., e:2 * e], y[..., 2 * e:]


# ---- fp32 Linears as ONE fp16 GEMM on operand triples (include/oeh.h: oeh_split_triples): x W^T + b to ~2^-22 relative - measured
# closer to the float64 result than the fp32 library GEMM - at about twice its speed.  Inference only (no autograd).
TRIPLE_GEMM = True


def triple_weights(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """[Wh ; Wl 2^-6 ; Wh 2^-6 ; bh ; bl 2^-6 ; 0 x6] (3K + 8, N) fp16 for a (N, K) fp32 weight: W = Wh + Wl 2^-11, b = bh + bl 2^-11."""
    with torch.no_grad():
        w = weight.detach().float()
        wh = w.clamp(-65504.0, 65504.0).half()
        wl = ((w - wh.float()) * 2048.0).clamp(-65504.0, 65504.0).half()
        n, k = w.shape
        out = torch.zeros((3 * k + 8, n), dtype=torch.float16, device=w.device)
        out[:k] = wh.t()
        out[k:2 * k] = (wl.float() * 0.015625).half().t()
        out[2 * k:3 * k] = (wh.float() * 0.015625).half().t()
        if bias is not None:
            b = bias.detach().float()
            bh = b.clamp(-65504.0, 65504.0).half()
            out[3 * k] = bh
            out[3 * k + 1] = ((b - bh.float()) * 2048.0 * 0.015625).clamp(-65504.0, 65504.0).half()
    return out


def triple_gemm_ok(x: torch.Tensor, *lins: nn.Linear) -> bool:
    return (TRIPLE_GEMM and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and all(type(m) is nn.Linear and m.weight.dtype == torch.float32 and m.in_features % 8 == 0 for m in lins) and not has_hooks(*lins))


def linear_fp32(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """`lin(x)` for an fp32 nn.Linear in inference: the triple GEMM when it applies (weights cached per module, rebuilt when a
    parameter changes), the module's own forward otherwise."""
    if not triple_gemm_ok(x, lin):
        return lin(x)
    key = (lin.weight._version, lin.weight.data_ptr(), None if lin.bias is None else (lin.bias._version, lin.bias.data_ptr()))
    cache = lin.__dict__.get("_oeh_triple_cache")
    if cache is None or cache[0] != key:
        cache = (key, triple_weights(lin.weight, lin.bias))
        lin.__dict__["_oeh_triple_cache"] = cache
    a = ops.split_triples(x.reshape(-1, x.shape[-1]))
    return torch.mm(a, cache[1], out_dtype=torch.float32).view(*x.shape[:-1], lin.out_features)


def has_hooks(*mods: nn.Module) -> bool:
    return any(len(m._forward_hooks) or len(m._forward_pre_hooks) for m in mods)


def split_mask(mask: Optional[torch.Tensor], B: int, Sq: int, Sk: int):
    """Classify an additive HF mask: (B,1,1,Sk) -> key padding vector; (B,1,Sq,Sk) -> full mask."""
    if mask is None:
        return None, None
    if mask.dim() == 4 and mask.shape[1] == 1 and mask.shape[2] == 1 and mask.shape[0] in (1, B) and mask.shape[3] == Sk:
        return mask.expand(B, 1, 1, Sk).reshape(B, Sk), None
    if mask.dim() == 4 and tuple(mask.shape) == (B, 1, Sq, Sk):
        return None, mask
    return None, mask.expand(B, 1, Sq, Sk)


_causal_cache = {}  # id(mask) -> (weakref to the mask tensor, its version counter, result)


def classify_causal(mask: torch.Tensor):
    """Recognise HF's decoder mask: a (B,1,T,S) additive tensor that equals causal(finfo.min above the shifted
    diagonal) + key-padding(finfo.min columns), up to the clamp at finfo.min the attention applies anyway
    (opt_attention.py:220-223).  Returns (True, pad_vector_or_None) or (False, None).

    One pass over the mask per distinct tensor OBJECT: HF hands the same tensor to every layer of a forward, so the
    result is remembered - keyed on the identity of the live object (a weak reference that must still point at
    `mask`) and its version counter, never on its address: the caching allocator gives the next batch's mask the
    same address, and a result remembered by address would apply the previous batch's padding to it."""
    key = id(mask)
    hit = _causal_cache.get(key)
    if hit is not None:
        if hit[0]() is mask and hit[1] == mask._version:
            return hit[2]
        del _causal_cache[key]
    B, _, T, S = mask.shape
    fmin = torch.finfo(mask.dtype).min
    causal = torch.full((T, S), fmin, dtype=mask.dtype, device=mask.device).triu(1 + S - T)
    pad = mask[:, 0, -1, :].clamp(min=fmin)  # the last query row sees every key: what is left is padding
    recon = (causal[None, None] + pad[:, None, None, :]).clamp(min=fmin)
    ok = bool(torch.equal(mask.clamp(min=fmin), recon)) and bool(((pad == 0) | (pad == fmin)).all())
    res = (True, (pad.contiguous() if bool((pad != 0).any()) else None)) if ok else (False, None)

    def _forget(dead, _k=key):  # the mask object is gone: its id may be reused by any other object
        ent = _causal_cache.get(_k)
        if ent is not None and ent[0] is dead:
            del _causal_cache[_k]

    ref = weakref.ref(mask, _forget)
    if len(_causal_cache) > 64:
        _causal_cache.clear()
    _causal_cache[key] = (ref, mask._version, res)
    return res


_padbool_cache = {}  # id(mask) -> (weakref to the mask tensor, its version counter, result)


def pad_is_boolean(mask: torch.Tensor) -> bool:
    """True when an additive mask holds only 0 (visible) and values <= -1e4 (hidden) - HF's extended masks hold 0 / finfo.min.
    The integer-matrix-core attention (`ops.attn_fwd_i8`) DROPS a padded key instead of adding the mask value to its score, which
    is the reference's result exactly for such masks.  One pass (and one host read) per distinct mask tensor object - HF hands
    the same tensor to every layer -, remembered like `classify_causal` does: by the identity of the live object and its
    version counter, never by address."""
    key = id(mask)
    hit = _padbool_cache.get(key)
    if hit is not None:
        if hit[0]() is mask and hit[1] == mask._version:
            return hit[2]
        del _padbool_cache[key]
    res = bool(((mask == 0) | (mask <= -1.0e4)).all())

    def _forget(dead, _k=key):
        ent = _padbool_cache.get(_k)
        if ent is not None and ent[0] is dead:
            del _padbool_cache[_k]

    if len(_padbool_cache) > 64:
        _padbool_cache.clear()
    _padbool_cache[key] = (weakref.ref(mask, _forget), mask._version, res)
    return res


def attention_core(
    q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, softmax_fn, scale: float = 1.0, scale_div: float = 0.0,
    attention_mask: Optional[torch.Tensor] = None, clamp_min: bool = False, detect_causal: bool = False,
    gate: Optional[torch.Tensor] = None, fq: Optional[AttnFakeQuant] = None, mask_min: Optional[float] = None,
    gate_mlp=None,
) -> torch.Tensor:
    """Fused path: logical (B,H,S,d) views in, (B,Sq,H*d) context out (head merge is free).  `gate_mlp`
    (ops.GatePredictor): the per-token gate is evaluated inside the attention kernel where the library supports that
    (full-row 16-bit kernel, <= 16 hidden units), else by `ops.gate_fwd` first."""
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    spec = spec_of(softmax_fn)
    pad, full = split_mask(attention_mask, B, Sq, Sk)
    causal = False
    if full is not None and detect_causal and Sq <= Sk:
        causal, padvec = classify_causal(full)
        if causal:
            full, pad = None, padvec
    if mask_min is None:
        mdt = attention_mask.dtype if attention_mask is not None and attention_mask.is_floating_point() else q.dtype
        mask_min = float(torch.finfo(mdt).min)
    # the fused INT8 chain stays on the quantiser grid with padded keys when the mask is a mask (0 / finfo.min entries - what
    # classify_causal has verified for a decoder mask; else one look per mask tensor object): include/oeh.h key_pad_boolean
    pad_bool = pad is not None and fq is not None and (causal or pad_is_boolean(attention_mask))
    kw = dict(softmax=spec, scale=scale, scale_div=scale_div, key_pad_mask=pad, full_mask=full, key_pad_boolean=pad_bool, causal=causal,
              clamp_min=clamp_min, mask_min=mask_min, fq=fq)
    out = None
    if gate_mlp is not None:
        units = 0 if gate_mlp.w1.dim() == 2 else gate_mlp.w1.shape[1]
        if full is None and ops.fused_gate_ok(B, H, Sq, Sk, D, q.dtype, clip=bool(spec.clip), fq=fq is not None, units=units,
                                              base=spec.base, gamma=spec.gamma, key_pad=pad is not None, causal=causal,
                                              scale=scale, scale_div=scale_div, mask_min=mask_min):
            try:
                out = ops.attn_fwd(q, k, v, gate_mlp=gate_mlp, **kw)
            except _lib.OehError as e:  # an option combination the 16-bit MFMA kernels do not take after all (alignment ...)
                if e.code not in (-95, -14):
                    raise
        if out is None:
            gp = gate_mlp
            g = ops.gate_fwd(gp.hidden, H, gp.w1, gp.b1, gp.w2, gp.b2, scaling=1.0)
            if gp.out is not None:
                gp.out.copy_(g[..., 0])
            gate = g * gp.scaling
    if out is None:
        out = ops.attn_fwd(q, k, v, gate=gate, **kw)
    return out.permute(0, 2, 1, 3).reshape(B, Sq, H * D)


def unfused_core(
    q, k, v, *, softmax_fn, scale: float = 1.0, scale_div: float = 0.0, attention_mask=None, clamp_min: bool = False,
    scores_tap=None, probs_tap=None, dropout=None, probs_after_tap=None, head_mask=None, extra_scores=None,
    fq_scores=None, fq_probs=None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Observable path (output_attentions, forward hooks on the Identity taps, head_mask, training dropout,
    relative position scores, user-supplied softmax callables): the (B,H,Sq,Sk) tensors exist, the two
    contractions are rocBLAS batched GEMMs and the softmax is the HIP row kernel behind SOFTMAX_MAPPING.
    Op order of bert_attention.py:222-292 / opt_attention.py:204-263.
    Returns (context (B,H,Sq,d), probs before dropout/head-mask, probs actually multiplied with V)."""
    ops._need_gpu(q, k, v, allow_grad=True)  # GPU only, like the fused path: this package has no CPU implementation; differentiable
    scores = torch.matmul(q, k.transpose(-1, -2))
    if extra_scores is not None:
        scores = scores + extra_scores
    if scale_div:
        scores = scores / scale_div
    elif scale != 1.0:
        scores = scores * scale
    if fq_scores is not None:
        scores = fq_scores(scores)
    if scores_tap is not None:
        scores = scores_tap(scores)
    if attention_mask is not None:
        scores = scores + attention_mask
        if clamp_min:
            scores = torch.max(scores, torch.tensor(torch.finfo(scores.dtype).min, device=scores.device))
    probs = softmax_fn(scores, dim=-1)
    if fq_probs is not None:
        probs = fq_probs(probs)
    if probs_tap is not None:
        probs = probs_tap(probs)
    used = probs
    if dropout is not None:
        used = dropout(used)
    if probs_after_tap is not None:
        used = probs_after_tap(used)
    if head_mask is not None:
        used = used * head_mask
    return torch.matmul(used, v), probs, used
