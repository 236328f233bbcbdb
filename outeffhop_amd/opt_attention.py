"""Drop-in `OPTAttentionWithExtras` (reference: OutEffHop/transformers_language/models/opt_attention.py:14-326).

Same constructor / forward signature, parameter names (q_proj, k_proj, v_proj, out_proj, alpha[...]) and
(attn_output, attn_weights_reshaped|None, past_key_value) result, so the reference's swap-in code
(validate_clm.py:147-168, run_clm.py:214-233) and checkpoints work unchanged.

Deviation (documented in DESIGN.md): with fp16 scores the reference calls softmax_fn(..., dtype=torch.float32)
(:227-230), which raises TypeError for the softmax_1 family; this module always does the softmax arithmetic in
fp32 inside the kernel - the semantics that branch intended - so OPT + softmax1 + fp16 runs.
"""
from __future__ import annotations

from functools import partial
from typing import Optional, Tuple

import torch
from torch import nn

from .attention import AttentionGateType, GateBookkeeping, GateState, attention_core, autograd_needed, build_gate, fused_qkv, has_hooks, linear_fp32, unfused_core
from .softmax import make_clipped_softmax, make_clipped_softmax1, spec_of


class OPTAttentionWithExtras(GateBookkeeping, nn.Module):
    """Multi-headed attention with the OutEffHop extras (modified softmax, gating)."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0, is_decoder: bool = False, bias: bool = True,
                 softmax_fn=torch.nn.functional.softmax, alpha=None, max_seq_length=None, ssm_eps=None, tau=None, skip_attn=False,
                 attn_gate_type=AttentionGateType.none, attn_gate_init=None, attn_gate_mlp=False, attn_gate_mlp2=False,
                 attn_gate_linear_all_features=False, fine_tuning=False, attn_softmax=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        if self.head_dim * num_heads != self.embed_dim:
            raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {self.embed_dim}"
                             f" and `num_heads`: {num_heads}).")
        self.scaling = self.head_dim ** -0.5
        self.is_decoder = is_decoder
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.v_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.attn_scores = nn.Identity()
        self.attn_probs_before_dropout = nn.Identity()
        self.attn_probs_after_dropout = nn.Identity()
        self.max_seq_length, self.ssm_eps, self.tau, self.attn_softmax = max_seq_length, ssm_eps, tau, attn_softmax
        if alpha is not None:  # opt_attention.py:70-77 (the reference tests `attn_softmax is "softmax1"`)
            assert max_seq_length is not None
            # = partial(clipped_softmax[1], gamma=, eta=1.0) as in the reference (a functools.partial subclass that also carries .spec)
            self.softmax_fn = (make_clipped_softmax1 if attn_softmax == "softmax1" else make_clipped_softmax)(-alpha / max_seq_length, 1.0)
        else:
            self.softmax_fn = softmax_fn
        self.skip_attn = skip_attn  # accepted and ignored, as in the reference
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
        self.alpha = build_gate(num_heads, self.head_dim, embed_dim, attn_gate_type, attn_gate_init, attn_gate_mlp, attn_gate_mlp2,
                                attn_gate_linear_all_features, fine_tuning, ft_std=0.001)

    def _heads(self, t: torch.Tensor, bsz: int) -> torch.Tensor:
        """(B,S,E) -> logical (B,H,S,d) view; no copy (the kernel takes strides)."""
        return t.view(bsz, -1, self.num_heads, self.head_dim).permute(0, 2, 1, 3)

    def _shape(self, tensor: torch.Tensor, seq_len: int, bsz: int):
        return tensor.view(bsz, seq_len, self.num_heads, self.head_dim).transpose(1, 2).contiguous()

    def forward(self, hidden_states: torch.Tensor, key_value_states: Optional[torch.Tensor] = None,
                past_key_value: Optional[Tuple[torch.Tensor]] = None, attention_mask: Optional[torch.Tensor] = None,
                layer_head_mask: Optional[torch.Tensor] = None, output_attentions: bool = False
                ) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[Tuple[torch.Tensor]]]:
        """Input shape: Batch x Time x Channel"""
        bsz, tgt_len, _ = hidden_states.size()
        qkv = None
        if key_value_states is None:  # self-attention: one GEMM for the three projections, the q scaling folded in (attention.fused_qkv)
            qkv = fused_qkv(self, hidden_states, self.q_proj, self.k_proj, self.v_proj, self.scaling)
        q = self._heads(qkv[0] if qkv is not None else self.q_proj(hidden_states) * self.scaling, bsz)
        if qkv is not None:
            k, v = self._heads(qkv[1], bsz), self._heads(qkv[2], bsz)
            if past_key_value is not None:
                k = torch.cat([past_key_value[0], k], dim=2)
                v = torch.cat([past_key_value[1], v], dim=2)
        elif key_value_states is not None and past_key_value is not None:
            k, v = past_key_value[0], past_key_value[1]
        elif key_value_states is not None:
            k, v = self._heads(self.k_proj(key_value_states), bsz), self._heads(self.v_proj(key_value_states), bsz)
        else:
            k, v = self._heads(self.k_proj(hidden_states), bsz), self._heads(self.v_proj(hidden_states), bsz)
            if past_key_value is not None:
                k = torch.cat([past_key_value[0], k], dim=2)
                v = torch.cat([past_key_value[1], v], dim=2)
        new_past = (k, v) if self.is_decoder else past_key_value
        src_len = k.size(2)
        if attention_mask is not None and attention_mask.size() != (bsz, 1, tgt_len, src_len):
            raise ValueError(f"Attention mask should be of size {(bsz, 1, tgt_len, src_len)}, but is {attention_mask.size()}")
        if layer_head_mask is not None and layer_head_mask.size() != (self.num_heads,):
            raise ValueError(f"Head mask for a single layer should be of size {(self.num_heads,)}, but is {layer_head_mask.size()}")
        fusable = (spec_of(self.softmax_fn) is not None and layer_head_mask is None and not output_attentions
                   and not (self.training and self.dropout > 0.0) and not autograd_needed(self, hidden_states, key_value_states, q, k, v)
                   and not has_hooks(self.attn_scores, self.attn_probs_before_dropout, self.attn_probs_after_dropout))
        # conditional per-token gate: evaluated inside the attention kernel when the fused path runs (self-attention only:
        # the predictor acts on the query-side layer input)
        gp = None
        if fusable and key_value_states is None and past_key_value is None:
            gp = GateState.predictor(self, hidden_states, self.num_heads, self.gate_scaling_factor)
        gate = None
        if gp is None:
            gate = GateState.evaluate(self, hidden_states, self.num_heads)
            if gate is not None and self.attn_gate_type != AttentionGateType.unconditional_per_head:
                gate = gate * self.gate_scaling_factor
        weights = None
        if fusable:
            merged = attention_core(q, k, v, softmax_fn=self.softmax_fn, scale=1.0, attention_mask=attention_mask,
                                    clamp_min=attention_mask is not None, detect_causal=True, gate=gate, gate_mlp=gp)
            if gp is not None:
                GateState.finish_predictor(self, gp, self.num_heads)
        else:
            hm = None if layer_head_mask is None else layer_head_mask.view(1, -1, 1, 1)
            drop = (lambda p: nn.functional.dropout(p, p=self.dropout, training=self.training))
            fn = self.softmax_fn
            if q.dtype == torch.float16 and spec_of(fn) is not None:  # upcast branch of the reference (:227-230), without its TypeError
                fn = (lambda x, dim=-1, _f=self.softmax_fn: _f(x.float(), dim=dim).to(torch.float16))
            ctx, _, used = unfused_core(q, k, v, softmax_fn=fn, attention_mask=attention_mask, clamp_min=attention_mask is not None,
                                        scores_tap=self.attn_scores, probs_tap=self.attn_probs_before_dropout, dropout=drop,
                                        probs_after_tap=self.attn_probs_after_dropout, head_mask=hm)
            weights = used if output_attentions else None
            if gate is not None:
                ctx = ctx * gate.to(ctx.dtype)
            merged = ctx.transpose(1, 2).reshape(bsz, tgt_len, self.embed_dim)
        return linear_fp32(self.out_proj, merged), weights, new_past  # (fp32 inference: one fp16 GEMM on operand triples)
