#!/usr/bin/env python3
"""Host time of the eager quantised BERT-base attention forward (95 us eager against 49 us of GPU work): cProfile, top by self time."""
import cProfile, os, pstats, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import outeffhop_amd as oa
from outeffhop_amd.bert_attention import BertSelfAttentionWithExtras
from outeffhop_amd.softmax import SOFTMAX_MAPPING

dev = torch.device("cuda:0")
torch.manual_seed(0)
fmin = torch.finfo(torch.float32).min
cfg = oa.get_quant_config()
cfg.act_quant.options = dict(percentile=99.999)
bcfg = SimpleNamespace(hidden_size=768, num_attention_heads=12, attention_probs_dropout_prob=0.0, max_position_embeddings=512, is_decoder=False,
                       position_embedding_type="absolute")
B, S, E = 32, 128, 768
with torch.no_grad():
    borg = BertSelfAttentionWithExtras(bcfg, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
    bq = oa.QuantizedBertSelfAttentionWithExtras(borg, **{**oa.val_qparams(cfg), "quant_dict": {}}).to(dev).eval()
    bq.set_quant_state(weight_quant=True, act_quant=True)
    mask = torch.zeros(B, 1, 1, S, device=dev)
    mask[:, :, :, 100:] = fmin
    for _ in range(2):
        bq(torch.randn(B, S, E, device=dev), attention_mask=mask)
    bq.fix_ranges()
    x = torch.randn(B, S, E, device=dev)
    for _ in range(50):
        bq(x, attention_mask=mask)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        bq(x, attention_mask=mask)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host time per forward {1e6 * (t1 - t0) / n:.1f} us")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        bq(x, attention_mask=mask)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(24)
