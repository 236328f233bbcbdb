"""Where the host time of an eager fp16 BERT-base / OPT-125m attention layer goes (cProfile over 3 000 forwards, GPU box)."""
import cProfile
import os
import pstats
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from outeffhop_amd import SOFTMAX_MAPPING, BertSelfAttentionWithExtras, OPTAttentionWithExtras

dev = torch.device("cuda:0")
dt = torch.float16
which = sys.argv[1] if len(sys.argv) > 1 else "bert"
with torch.no_grad():
    if which == "bert":
        cfg = SimpleNamespace(hidden_size=768, num_attention_heads=12, attention_probs_dropout_prob=0.0, max_position_embeddings=512, is_decoder=False,
                              position_embedding_type="absolute")
        B, S, E = 32, 128, 768
        m = BertSelfAttentionWithExtras(cfg, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).to(dt).eval()
        x = torch.randn(B, S, E, device=dev, dtype=dt)
        pad = torch.zeros(B, 1, 1, S, device=dev, dtype=dt)
        pad[:, :, :, 100:] = torch.finfo(dt).min
        fn = lambda: m(x, attention_mask=pad)  # noqa: E731
    else:
        B, S, E, H = 16, 512, 768, 12
        m = OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).to(dt).eval()
        x = torch.randn(B, S, E, device=dev, dtype=dt)
        fmin = torch.finfo(torch.float32).min
        mask = torch.clamp(torch.full((S, S), fmin, device=dev).triu(1)[None, None].expand(B, 1, S, S), min=torch.finfo(dt).min).to(dt)
        fn = lambda: m(x, attention_mask=mask)  # noqa: E731
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3000):
        fn()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
