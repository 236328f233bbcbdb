"""Hopfield association layers of the STanHop / theory_verification trees on the fused kernel.

Reference: STanHop_time_seeries/cross_models/hopfield.py:19-141 (Association, Hopfield, HopfieldPooling) and
theory_verification/layers.py:90-178.  Layout (B,L,H,E) / (B,S,H,E) / (B,S,H,D) -> (B,L,H,D); L may differ from S.

Modes: 'softmax1' | 'softmax' | 'clip' | 'clip_softmax1' run on the GPU kernel.  The reference's
ClipSoftmax_1.__init__ calls super(ClipSoftmax, self) (clip_softmax.py:46) so mode='clip_softmax1' raises TypeError
there; here it works.  'entmax' (the constructor default: alpha-entmax with a learnable alpha) and 'sparsemax' are sort /
bisection based activations outside the HIP hot path: torch ops (sparse_activations.py) around two library GEMMs.
"""
from __future__ import annotations

from math import sqrt

import torch
from torch import nn

from .attention import unfused_core
from .ops import SoftmaxSpec, attn_fwd, grad_recording
from .softmax import SoftmaxFn
from .sparse_activations import EntmaxAlpha, Sparsemax

_MODES = {
    "softmax1": lambda eta, gamma: SoftmaxSpec(1, False, 0.0, 1.0),
    "softmax": lambda eta, gamma: SoftmaxSpec(0, False, 0.0, 1.0),
    "clip": lambda eta, gamma: SoftmaxSpec(0, True, gamma, eta),
    "clip_softmax1": lambda eta, gamma: SoftmaxSpec(1, True, gamma, eta),
}


class Association(nn.Module):
    def __init__(self, scale=None, attention_dropout=0.1, eta=1.1, gamma=-0.1, mode="entmax"):
        super().__init__()
        self.scale = scale
        self.dropout = nn.Dropout(attention_dropout)
        self.mode = mode
        if mode in _MODES:
            self.softmax = SoftmaxFn(mode, _MODES[mode](eta, gamma))
        elif mode == "entmax":
            self.softmax = EntmaxAlpha()
        elif mode == "sparsemax":
            self.softmax = Sparsemax()
        else:
            raise ValueError(f"Association mode {mode!r}: one of {sorted(_MODES) + ['entmax', 'sparsemax']}")

    def forward(self, queries, keys, values):
        B, L, H, E = queries.shape
        scale = self.scale or 1.0 / sqrt(E)
        q, k, v = queries.permute(0, 2, 1, 3), keys.permute(0, 2, 1, 3), values.permute(0, 2, 1, 3)
        if not isinstance(self.softmax, SoftmaxFn):  # sparse activations: scores materialised, torch ops (outside the HIP path)
            probs = self.dropout(self.softmax(scale * torch.matmul(q, k.transpose(-1, -2))))
            return torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
        if (self.training and self.dropout.p > 0.0) or grad_recording(q, k, v):  # dropout / autograd: the observable torch-op path
            ctx, _, _ = unfused_core(q, k, v, softmax_fn=self.softmax, scale=scale, dropout=self.dropout)
            return ctx.permute(0, 2, 1, 3).contiguous()
        out = attn_fwd(q, k, v, softmax=self.softmax.spec, scale=scale)  # stored (B,L,H,D)-contiguous
        return out.permute(0, 2, 1, 3)


class Hopfield(nn.Module):
    """Multi-head Hopfield layer; V is projected from the PROJECTED keys (hopfield.py:78)."""

    def __init__(self, d_model, n_heads, d_keys=None, d_values=None, mix=True, dropout=0.1, eta=1.1, gamma=-0.1, mode="entmax"):
        super().__init__()
        d_keys = d_keys or (d_model // n_heads)
        d_values = d_values or (d_model // n_heads)
        self.inner_attention = Association(scale=None, attention_dropout=dropout, eta=eta, gamma=gamma, mode=mode)
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_model, d_keys * n_heads)
        self.value_projection = nn.Linear(d_keys * n_heads, d_values * n_heads)
        self.out_projection = nn.Linear(d_values * n_heads, d_model)
        self.n_heads = n_heads
        self.mix = mix

    def forward(self, queries, keys, values):
        B, L, _ = queries.shape
        S = keys.shape[1]
        H = self.n_heads
        q = self.query_projection(queries).view(B, L, H, -1)
        kp = self.key_projection(keys)
        v = self.value_projection(kp).view(B, S, H, -1)
        out = self.inner_attention(q, kp.view(B, S, H, -1), v)
        if self.mix:
            out = out.transpose(2, 1).contiguous()
        return self.out_projection(out.reshape(B, L, -1))


class HopfieldPooling(nn.Module):
    """Hopfield layer whose keys are `num_pattern` learned patterns (hopfield.py:92-141)."""

    def __init__(self, d_model, n_heads, num_pattern=1, d_keys=None, d_values=None, mix=True, dropout=0.1, eta=1.1, gamma=-0.1,
                 mode="entmax"):
        super().__init__()
        d_keys = d_keys or (d_model // n_heads)
        d_values = d_values or (d_model // n_heads)
        self.inner_attention = Association(scale=None, attention_dropout=dropout, eta=eta, gamma=gamma, mode=mode)
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_model, d_keys * n_heads)
        self.value_projection = nn.Linear(d_keys * n_heads, d_values * n_heads)
        self.out_projection = nn.Linear(d_values * n_heads, d_model)
        self.n_heads = n_heads
        self.mix = mix
        self.key = nn.Parameter(torch.empty(1, num_pattern, d_model), requires_grad=True)

    def forward(self, query):
        B, L, _ = query.shape
        S = self.key.shape[1]
        H = self.n_heads
        q = self.query_projection(query).view(B, L, H, -1)
        kp = self.key_projection(self.key.repeat(B, 1, 1))
        v = self.value_projection(kp).view(B, S, H, -1)
        out = self.inner_attention(q, kp.view(B, S, H, -1), v)
        if self.mix:
            out = out.transpose(2, 1).contiguous()
        return self.out_projection(out.reshape(B, L, -1))
