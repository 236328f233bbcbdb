"""Drop-in `ViTSelfAttentionWithExtras` (reference: OutEffHop/transformers_language/models/vit_attention.py:77-269)
for timm-style ViT blocks: fused qkv projection, optional q/k LayerNorm, modified softmax, gating, output projection.
forward(x) -> Tensor.  N = 197 tokens is not a multiple of 16: the kernel masks the key tail.

The reference's `attn_gate_linear_all_features=True` path reads an undefined `self.all_head_size` (:162); here it is
defined (= dim) so the option works instead of raising AttributeError.
"""
from __future__ import annotations

from functools import partial

import torch
from torch import nn

from .attention import AttentionGateType, GateBookkeeping, GateState, attention_core, autograd_needed, build_gate, has_hooks, linear_fp32, unfused_core
from .softmax import spec_of


class ViTSelfAttentionWithExtras(GateBookkeeping, nn.Module):
    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, qk_norm: bool = False, attn_drop: float = 0.0,
                 proj_drop: float = 0.0, norm_layer: nn.Module = nn.LayerNorm, softmax_fn=torch.nn.functional.softmax, gamma=None,
                 ssm_eps=None, tau=None, skip_attn=False, attn_gate_type=AttentionGateType.none, attn_gate_init=None,
                 attn_gate_mlp=False, attn_gate_mlp2=False, attn_gate_linear_all_features=False, fine_tuning=False,
                 max_seq_length=None) -> None:
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_attention_heads = num_heads
        self.attention_head_size = dim // num_heads
        self.all_head_size = dim
        self.scale = self.attention_head_size ** -0.5
        self.fused_attn = True
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = norm_layer(self.attention_head_size) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.attention_head_size) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.attn_scores = nn.Identity()
        self.attn_probs_before_dropout = nn.Identity()
        self.attn_probs_after_dropout = nn.Identity()
        self.gamma, self.ssm_eps, self.tau, self.max_seq_length = gamma, ssm_eps, tau, max_seq_length
        self.softmax_fn = softmax_fn
        self.skip_attn = skip_attn
        self.last_gate_avg_prob = None
        self.last_gate_all_probs = None
        self.attn_gate_type = attn_gate_type
        self.attn_gate_init = attn_gate_init
        self.attn_gate_mlp = attn_gate_mlp
        self.attn_gate_mlp2 = attn_gate_mlp2
        self.attn_gate_linear_all_features = attn_gate_linear_all_features
        self.gate_fn = torch.sigmoid
        self.pooling_fn = partial(torch.mean, dim=1, keepdims=True)
        self.fine_tuning = fine_tuning
        self.gate_scaling_factor = 1.0 / attn_gate_init if (fine_tuning and attn_gate_init is not None) else 1.0
        self.alpha = build_gate(num_heads, self.attention_head_size, dim, attn_gate_type, attn_gate_init, attn_gate_mlp,
                                attn_gate_mlp2, attn_gate_linear_all_features, fine_tuning, ft_std=0.01)

    def transpose_for_scores(self, x: torch.Tensor) -> torch.Tensor:
        return x.view(x.size()[:-1] + (self.num_attention_heads, self.attention_head_size)).permute(0, 2, 1, 3)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, N, C = x.shape
        H, d = self.num_attention_heads, self.attention_head_size
        q, k, v = linear_fp32(self.qkv, x).reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4).unbind(0)  # (B,H,N,d) views, unit d stride
        q, k = self.q_norm(q), self.k_norm(k)
        fusable = (spec_of(self.softmax_fn) is not None and not (self.training and self.attn_drop.p > 0.0) and not autograd_needed(self, x, q, k, v)
                   and not has_hooks(self.attn_scores, self.attn_probs_before_dropout, self.attn_probs_after_dropout))
        gp = GateState.predictor(self, x, H, self.gate_scaling_factor) if fusable else None  # gate evaluated in the kernel
        gate = None
        if gp is None:
            gate = GateState.evaluate(self, x, H)
            if gate is not None and self.attn_gate_type != AttentionGateType.unconditional_per_head:
                gate = gate * self.gate_scaling_factor
        if fusable:
            merged = attention_core(q, k, v, softmax_fn=self.softmax_fn, scale=self.scale, gate=gate, gate_mlp=gp)
            if gp is not None:
                GateState.finish_predictor(self, gp, H)
        else:
            ctx, _, _ = unfused_core(q, k, v, softmax_fn=self.softmax_fn, scale=self.scale, scores_tap=self.attn_scores,
                                     probs_tap=self.attn_probs_before_dropout, dropout=self.attn_drop,
                                     probs_after_tap=self.attn_probs_after_dropout)
            if gate is not None:
                ctx = ctx * gate.to(ctx.dtype)
            merged = ctx.transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(linear_fp32(self.proj, merged))
