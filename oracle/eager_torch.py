"""Eager-PyTorch restatement of the reference's attention op chain (CPU baseline + second checker).

TEST / BASELINE INFRASTRUCTURE ONLY (same rule as oeh_oracle.py): imported by tests/ and by bench.py's
`cpu_baseline` leg, never by the product package.

The reference IS a chain of eager torch ops, so the faithful "reference CPU path" that can travel to the GPU
box (the reference's Python cannot) is the same torch ops in the same order:
    OPT  : opt_attention.py:204 (bmm) :220-224 (+mask, max(., finfo.min)) :232 (softmax_fn) :263 (bmm)
    BERT : bert_attention.py:222 (matmul) :265 (/sqrt(d)) :272 (+mask) :276 (softmax_fn) :292 (matmul)
    softmax_1 : vutils/softmax_1.py:11-21 ; clip : models/softmax.py:18-19
    fake-quant: uniform_quantizers.py:114-115,146
Parity status: pinned - tests/test_oracle_golden.py::test_eager_torch_matches_golden checks it against the
fixtures captured from the reference (bit-exact elementwise chain, BLAS-order tolerance on matmuls).
"""
from __future__ import annotations

import math
from typing import Optional

import torch


def softmax_n_shifted_zeros(x: torch.Tensor, n: float, dim: int = -1) -> torch.Tensor:
    m = x.max(dim=dim, keepdim=True).values  # softmax_1.py:11
    e = torch.exp(torch.subtract(x, m))  # :13-15
    s = e.sum(dim=dim, keepdim=True)  # :16
    den = torch.add(s, torch.multiply(torch.exp(torch.multiply(m, -1)), n))  # :18-20
    return torch.divide(e, den)  # :21


def softmax_fn(x: torch.Tensor, base: int, clip: bool, gamma: float, eta: float) -> torch.Tensor:
    p = softmax_n_shifted_zeros(x, 1, -1) if base == 1 else torch.nn.functional.softmax(x, dim=-1)
    if clip:
        p = torch.clip(p * (eta - gamma) + gamma, 0, 1)  # softmax.py:12-13 / :18-19
    return p


def fake_quant(x: torch.Tensor, scale: float, zp: float, qmax: float) -> torch.Tensor:
    xi = torch.clamp(torch.round(x / scale) + zp, 0.0, qmax)  # uniform_quantizers.py:114-115
    return scale * (xi - zp)  # :146


def attn_core_eager(q, k, v, *, order: str = "opt", base: int = 1, clip: bool = False, gamma: float = 0.0, eta: float = 1.0,
                    mask: Optional[torch.Tensor] = None, fq=None, gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q,k,v (B,H,S,d) fp32 CPU tensors.  order "opt": q is already scaled, mask (B,1,T,S) + clamp;
    order "bert": scores / sqrt(d), mask (B,1,1,S).  fq = dict(scores=(s,z,qmax), probs=..., ctx=...)."""
    B, H, S, d = q.shape
    fq = fq or {}
    if order == "opt":
        w = torch.bmm(q.reshape(B * H, S, d), k.reshape(B * H, -1, d).transpose(1, 2))
        if "scores" in fq:
            w = fake_quant(w, *fq["scores"])
        if mask is not None:
            w = w.view(B, H, S, -1) + mask
            w = torch.max(w, torch.tensor(torch.finfo(w.dtype).min))
            w = w.view(B * H, S, -1)
        w = softmax_fn(w, base, clip, gamma, eta)
        if "probs" in fq:
            w = fake_quant(w, *fq["probs"])
        o = torch.bmm(w, v.reshape(B * H, -1, d))
        if "ctx" in fq:
            o = fake_quant(o, *fq["ctx"])
        o = o.view(B, H, S, d)
        if gate is not None:
            o = o * gate
        return o
    s = torch.matmul(q, k.transpose(-1, -2))
    s = s / math.sqrt(d)
    if "scores" in fq:
        s = fake_quant(s, *fq["scores"])
    if mask is not None:
        s = s + mask
    p = softmax_fn(s, base, clip, gamma, eta)
    if "probs" in fq:
        p = fake_quant(p, *fq["probs"])
    o = torch.matmul(p, v)
    if gate is not None:
        o = o * gate
    if "ctx" in fq:
        o = fake_quant(o, *fq["ctx"])
    return o


def causal_mask(B: int, T: int, dtype=torch.float32) -> torch.Tensor:
    """HF 4.31 `_make_causal_mask` semantics: finfo.min strictly above the diagonal, (B,1,T,T)."""
    fmin = torch.finfo(dtype).min
    return torch.full((T, T), fmin, dtype=dtype).triu(1)[None, None].expand(B, 1, T, T).contiguous()
