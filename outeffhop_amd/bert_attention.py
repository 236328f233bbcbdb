"""Drop-in `BertSelfAttentionWithExtras` (reference: OutEffHop/transformers_language/models/bert_attention.py:28-343).

Same constructor / forward signature, parameter names (query, key, value, alpha[...]) and output tuple, so the
reference's swap-in code (validate_mlm_config.py:174-191, run_mlm.py:200-219) and checkpoints work unchanged.
forward() runs the fused HIP attention kernel; the (B,H,S,S) tensors only exist when something asks to see them
(output_attentions, hooks on the Identity taps, head_mask, training dropout, relative position embeddings).
"""
from __future__ import annotations

import math
from functools import partial
from typing import Optional, Tuple

import torch
from torch import nn

from .attention import AttentionGateType, GateBookkeeping, GateState, attention_core, autograd_needed, build_gate, fused_qkv, has_hooks, unfused_core
from .softmax import make_clipped_softmax, spec_of


class BertSelfAttentionWithExtras(GateBookkeeping, nn.Module):
    def __init__(self, config, position_embedding_type=None, softmax_fn=torch.nn.functional.softmax, alpha=None, ssm_eps=None,
                 tau=None, max_seq_length=None, skip_attn=False, attn_gate_type=AttentionGateType.none, attn_gate_init=None,
                 attn_gate_mlp=False, attn_gate_mlp2=False, attn_gate_linear_all_features=False, fine_tuning=False):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0 and not hasattr(config, "embedding_size"):
            raise ValueError(f"The hidden size ({config.hidden_size}) is not a multiple of the number of attention "
                             f"heads ({config.num_attention_heads})")
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self.position_embedding_type = position_embedding_type or getattr(config, "position_embedding_type", "absolute")
        if self.position_embedding_type in ("relative_key", "relative_key_query"):
            self.max_position_embeddings = config.max_position_embeddings
            self.distance_embedding = nn.Embedding(2 * config.max_position_embeddings - 1, self.attention_head_size)
        self.is_decoder = config.is_decoder
        # observation taps (hooks on them force the unfused path)
        self.attn_scores = nn.Identity()
        self.attn_probs_before_dropout = nn.Identity()
        self.attn_probs_after_dropout = nn.Identity()
        self.ssm_eps, self.tau, self.max_seq_length = ssm_eps, tau, max_seq_length
        if alpha is not None:  # bert_attention.py:89-92: alpha selects a clipped softmax with gamma = -alpha / max_seq_length
            assert max_seq_length is not None
            self.softmax_fn = make_clipped_softmax(-alpha / max_seq_length, 1.0)  # = partial(clipped_softmax, gamma=, eta=1.0)
        else:
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
        self.alpha = build_gate(self.num_attention_heads, self.attention_head_size, self.all_head_size, attn_gate_type, attn_gate_init,
                                attn_gate_mlp, attn_gate_mlp2, attn_gate_linear_all_features, fine_tuning, ft_std=0.01)

    def transpose_for_scores(self, x: torch.Tensor) -> torch.Tensor:
        return x.view(x.size()[:-1] + (self.num_attention_heads, self.attention_head_size)).permute(0, 2, 1, 3)

    # ------------------------------------------------------------------------------------------------
    def _project(self, hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask, past_key_value):
        if encoder_hidden_states is None:  # self-attention: one GEMM for the three projections (attention.fused_qkv)
            qkv = fused_qkv(self, hidden_states, self.query, self.key, self.value)
            if qkv is not None:
                q, k, v = (self.transpose_for_scores(t) for t in qkv)
                if past_key_value is not None:
                    k = torch.cat([past_key_value[0], k], dim=2)
                    v = torch.cat([past_key_value[1], v], dim=2)
                return q, k, v, attention_mask
        q = self.transpose_for_scores(self.query(hidden_states))
        if encoder_hidden_states is not None:
            attention_mask = encoder_attention_mask
            if past_key_value is not None:
                k, v = past_key_value[0], past_key_value[1]
            else:
                k = self.transpose_for_scores(self.key(encoder_hidden_states))
                v = self.transpose_for_scores(self.value(encoder_hidden_states))
        else:
            k = self.transpose_for_scores(self.key(hidden_states))
            v = self.transpose_for_scores(self.value(hidden_states))
            if past_key_value is not None:
                k = torch.cat([past_key_value[0], k], dim=2)
                v = torch.cat([past_key_value[1], v], dim=2)
        return q, k, v, attention_mask

    def _relative_scores(self, q, k, use_cache, device):
        lq, lk = q.shape[2], k.shape[2]
        if use_cache:
            pos_l = torch.tensor(lk - 1, dtype=torch.long, device=device).view(-1, 1)
        else:
            pos_l = torch.arange(lq, dtype=torch.long, device=device).view(-1, 1)
        pos_r = torch.arange(lk, dtype=torch.long, device=device).view(1, -1)
        emb = self.distance_embedding(pos_l - pos_r + self.max_position_embeddings - 1).to(dtype=q.dtype)
        rel = torch.einsum("bhld,lrd->bhlr", q, emb)
        if self.position_embedding_type == "relative_key_query":
            rel = rel + torch.einsum("bhrd,lrd->bhlr", k, emb)
        return rel

    def _fusable(self, head_mask, output_attentions, *inputs) -> bool:
        return (spec_of(self.softmax_fn) is not None and head_mask is None and not output_attentions and not autograd_needed(self, *inputs)
                and not (self.training and self.dropout.p > 0.0) and self.position_embedding_type == "absolute"
                and not has_hooks(self.attn_scores, self.attn_probs_before_dropout, self.attn_probs_after_dropout))

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.FloatTensor] = None,
                head_mask: Optional[torch.FloatTensor] = None, encoder_hidden_states: Optional[torch.FloatTensor] = None,
                encoder_attention_mask: Optional[torch.FloatTensor] = None,
                past_key_value: Optional[Tuple[Tuple[torch.FloatTensor]]] = None,
                output_attentions: Optional[bool] = False) -> Tuple[torch.Tensor]:
        if self.skip_attn:
            return (torch.zeros_like(hidden_states),)
        use_cache = past_key_value is not None
        q, k, v, attention_mask = self._project(hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask, past_key_value)
        new_past = (k, v) if self.is_decoder else None
        fusable = self._fusable(head_mask, output_attentions, hidden_states, encoder_hidden_states, q, k, v)
        # conditional per-token gate: evaluated inside the attention kernel when the fused path runs
        gp = GateState.predictor(self, hidden_states, self.num_attention_heads, self.gate_scaling_factor) if fusable else None
        gate = None
        if gp is None:
            gate = GateState.evaluate(self, hidden_states, self.num_attention_heads)
            if gate is not None and self.attn_gate_type != AttentionGateType.unconditional_per_head:
                gate = gate * self.gate_scaling_factor  # context *= gate * scaling (bert_attention.py:327)
        div = math.sqrt(self.attention_head_size)
        probs = None
        if fusable:
            context = attention_core(q, k, v, softmax_fn=self.softmax_fn, scale_div=div, attention_mask=attention_mask, gate=gate,
                                     gate_mlp=gp)
            if gp is not None:
                GateState.finish_predictor(self, gp, self.num_attention_heads)
        else:
            extra = None
            if self.position_embedding_type in ("relative_key", "relative_key_query"):
                extra = self._relative_scores(q, k, use_cache, hidden_states.device)
            ctx, _, probs = unfused_core(q, k, v, softmax_fn=self.softmax_fn, scale_div=div, attention_mask=attention_mask,
                                         scores_tap=self.attn_scores, probs_tap=self.attn_probs_before_dropout, dropout=self.dropout,
                                         probs_after_tap=self.attn_probs_after_dropout, head_mask=head_mask, extra_scores=extra)
            if gate is not None:
                ctx = ctx * gate.to(ctx.dtype)
            context = ctx.permute(0, 2, 1, 3).contiguous().view(ctx.shape[0], ctx.shape[2], self.all_head_size)
        outputs = (context, probs) if output_attentions else (context,)
        if self.is_decoder:
            outputs = outputs + (new_past,)
        return outputs
