"""Which kernels a captured quantised BERT layer replays (torch profiler), against the eager plan path and the eager full path."""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import outeffhop_amd as oa
from outeffhop_amd import SOFTMAX_MAPPING, BertSelfAttentionWithExtras
from outeffhop_amd import quantization as Q

dev = torch.device("cuda:0")
torch.manual_seed(0)
E = 768
cfg = oa.get_quant_config()
cfg.quant.act_quant_method = "running_minmax" if not hasattr(cfg.quant, "act_quant_method") else cfg.quant.act_quant_method
fmin = torch.finfo(torch.float32).min


def kernels(fn, n=10):
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    rows = [(e.key[:90], e.count / n, e.device_time_total / n if hasattr(e, "device_time_total") else e.cuda_time_total / n) for e in prof.key_averages()]
    rows.sort(key=lambda r: -r[2])
    for k, c, t in rows[:12]:
        print(f"   {t:8.1f} us  x{c:4.1f}  {k}")
    print(f"   total {sum(r[2] for r in rows):8.1f} us per forward")


with torch.no_grad():
    bcfg = SimpleNamespace(hidden_size=768, num_attention_heads=12, attention_probs_dropout_prob=0.0, max_position_embeddings=512, is_decoder=False,
                           position_embedding_type="absolute")
    Bb, Sb = 32, 128
    borg = BertSelfAttentionWithExtras(bcfg, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
    bq = oa.QuantizedBertSelfAttentionWithExtras(borg, **{**oa.val_qparams(cfg), "quant_dict": {}}).to(dev).eval()
    bq.set_quant_state(weight_quant=True, act_quant=True)
    lens = torch.randint(Sb // 2, Sb + 1, (Bb,))
    bmask = torch.zeros(Bb, 1, 1, Sb, device=dev)
    for b_, n_ in enumerate(lens.tolist()):
        bmask[b_, :, :, n_:] = fmin
    for _ in range(2):
        bq(torch.randn(Bb, Sb, E, device=dev), attention_mask=bmask)
    bq.fix_ranges()
    xb = torch.randn(Bb, Sb, E, device=dev)
    bq(xb, attention_mask=bmask)
    bq(xb, attention_mask=bmask)
    print("eager (plan):", bq.__dict__.get("_i8_plan_runs"))
    kernels(lambda: bq(xb, attention_mask=bmask))
    Q.I8_PLAN = False
    print("eager (full path):")
    kernels(lambda: bq(xb, attention_mask=bmask))
    Q.I8_PLAN = True
    bq(xb, attention_mask=bmask)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out_b = bq(xb, attention_mask=bmask)
    print("captured graph replay:")
    kernels(gr.replay)
