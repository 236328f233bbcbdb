#!/usr/bin/env python3
"""Module-level timing (GPU box): the whole attention module (projections + fused core + output projection) of the OPT-125m
and BERT-base layers, and its parts, in attention-layer tokens/s.  SURVEY 8(f)-1: what sits either side of the core.
usage: python tools/module_bench.py [fp16|fp32]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from outeffhop_amd import attention
from outeffhop_amd.bert_attention import BertSelfAttentionWithExtras
from outeffhop_amd.opt_attention import OPTAttentionWithExtras
from outeffhop_amd.softmax import SOFTMAX_MAPPING


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.2:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.float16
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fmin = torch.finfo(torch.float32).min
    with torch.no_grad():
        # OPT-125m layer: B=16, S=512, E=768, H=12, causal mask tensor as HF passes it
        B, S, E, H = 16, 512, 768, 12
        m = OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).to(dt).eval()
        x = torch.randn(B, S, E, device=dev, dtype=dt)
        mask = torch.full((S, S), fmin, device=dev).triu(1)[None, None].expand(B, 1, S, S).to(dt if dt == torch.float32 else torch.float16)
        mask = torch.clamp(mask.float(), min=torch.finfo(dt).min).to(dt)
        for fused in (False, True):
            attention.FUSE_QKV = fused
            t_mod = timeit(lambda: m(x, attention_mask=mask))
            print(f"OPT-125m module {dt} fused_qkv={fused}: {t_mod:8.1f} us  {B * S / t_mod:8.1f} M tokens/s")
        t_q = timeit(lambda: m.q_proj(x))
        t_o = timeit(lambda: m.out_proj(x))
        print(f"   parts: one E->E Linear {t_q:.1f} us, out_proj {t_o:.1f} us")
        # BERT-base layer: B=32, S=128
        from types import SimpleNamespace
        cfg = SimpleNamespace(hidden_size=768, num_attention_heads=12, attention_probs_dropout_prob=0.0, max_position_embeddings=512,
                              is_decoder=False, position_embedding_type="absolute")
        B, S = 32, 128
        bm = BertSelfAttentionWithExtras(cfg, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).to(dt).eval()
        xb = torch.randn(B, S, E, device=dev, dtype=dt)
        pad = torch.zeros(B, 1, 1, S, device=dev, dtype=dt)
        pad[:, :, :, 100:] = torch.finfo(dt).min
        for fused in (False, True):
            attention.FUSE_QKV = fused
            t_mod = timeit(lambda: bm(xb, attention_mask=pad))
            print(f"BERT-base module {dt} fused_qkv={fused}: {t_mod:8.1f} us  {B * S / t_mod:8.1f} M tokens/s")
        # the same forwards replayed from a captured HIP graph: the modules launch on the current stream and never read the device
        # on the host (the OPT path classifies a mask tensor once, before the capture), so a whole forward can be captured
        for name, fn, ntok in (("BERT-base", lambda: bm(xb, attention_mask=pad), 32 * 128), ("OPT-125m", lambda: m(x, attention_mask=mask), 16 * 512)):
            fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                out_g = fn()
            t_g = timeit(gr.replay)
            ref = fn()
            same = torch.equal(ref[0], out_g[0])
            print(f"{name} module {dt} as a captured HIP graph: {t_g:8.1f} us  {ntok / t_g:8.1f} M tokens/s  (replay equals eager: {same})")


def quantised():
    """The INT8 validate configuration of the OPT-125m layer at the reference's precision (fp32 model): QuantizedOPTAttention-
    WithExtras with frozen 8-bit ranges, attention core on the integer matrix cores (INT8 storage) vs the fake-quant kernels."""
    import outeffhop_amd as oa
    from outeffhop_amd import quantization as Q

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fmin = torch.finfo(torch.float32).min
    B, S, E, H = 16, 512, 768, 12
    cfg = oa.get_quant_config()
    # (validate_clm.py:450-454: --percentile sets the ACTIVATION estimators' option only; round 5: this script also set cfg.quant.percentile, which
    # puts the percentile on the weights' CurrentMinMaxEstimator - the reference's (percentile, 100 - percentile) order there gives a degenerate,
    # unsigned weight grid - timings unaffected, but not the reference's configuration)
    cfg.act_quant.options = dict(percentile=99.999)
    with torch.no_grad():
        org = OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
        qm = oa.QuantizedOPTAttentionWithExtras(org, **{**oa.val_qparams(cfg), "quant_dict": {}}).to(dev).eval()
        qm.set_quant_state(weight_quant=True, act_quant=True)
        mask = torch.full((S, S), fmin, device=dev).triu(1)[None, None].expand(B, 1, S, S).contiguous()
        t0 = time.perf_counter()
        for _ in range(2):
            qm(torch.randn(B, S, E, device=dev), attention_mask=mask)  # calibration batches (percentile 99.999 + running average, on the device)
        torch.cuda.synchronize()
        print(f"calibration: 2 batches in {(time.perf_counter() - t0) * 1e3:.1f} ms (incl. first-call overheads)")
        for fused in (True, False):  # estimate_ranges: oeh_attn_calibrate (no (B,H,S,S) tensors) vs the observable path (materialised)
            Q.FUSED_CALIBRATION = fused
            xs = [torch.randn(B, S, E, device=dev) for _ in range(3)]
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            t0 = time.perf_counter()
            for x_ in xs:
                qm(x_, attention_mask=mask)
            torch.cuda.synchronize()
            print(f"calibration, {'in-kernel statistics (oeh_attn_calibrate)' if fused else 'materialised score tensors':42s}: "
                  f"{(time.perf_counter() - t0) * 1e3 / 3:.2f} ms per batch, peak extra memory {(torch.cuda.max_memory_allocated() - base) / 1e6:.0f} MB"
                  f" = {(torch.cuda.max_memory_allocated() - base) / (B * S * E * 4):.1f} (B,S,E) fp32 tensors of {B * S * E * 4 / 1e6:.0f} MB (projection outputs, their"
                  f" fake-quantised copies, the context); a (B,H,S,S) fp32 tensor would be {B * 12 * S * S * 4 / 1e6:.0f} MB")
        Q.FUSED_CALIBRATION = True
        qm.fix_ranges()
        x = torch.randn(B, S, E, device=dev)
        for i8, ig in ((False, False), (True, False), (True, True)):
            Q.INT8_STORAGE, Q.INDEX_GEMM = i8, ig
            t_mod = timeit(lambda: qm(x, attention_mask=mask), n=50)
            print(f"QuantizedOPT module fp32, INT8 storage core={i8}, out_proj on the context integers={ig}: {t_mod:8.1f} us  {B * S / t_mod:8.1f} M tokens/s")
        Q.INT8_STORAGE, Q.INDEX_GEMM = True, True
        try:  # the frozen-range INT8 forward as a captured HIP graph
            qm(x, attention_mask=mask)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                out_g = qm(x, attention_mask=mask)
            t_g = timeit(gr.replay, n=50)
            print(f"QuantizedOPT module fp32, INT8 storage core, captured HIP graph: {t_g:8.1f} us  {B * S / t_g:8.1f} M tokens/s  "
                  f"(replay equals eager: {torch.equal(out_g[0], qm(x, attention_mask=mask)[0])})")
        except Exception as e:  # noqa: BLE001
            print("capture of the quantised module failed:", type(e).__name__, str(e)[:200])
        t_lin = timeit(lambda: qm.q_proj(x), n=50)
        print(f"   parts: one QuantLinear (fp32 GEMM + output fake-quant) {t_lin:.1f} us")
        # BERT-base layer (quantized_bert.py:221-440): B = 32, S = 128, key-padding mask, query / key / value QuantLinear + the integer core
        from types import SimpleNamespace
        bcfg = SimpleNamespace(hidden_size=768, num_attention_heads=12, attention_probs_dropout_prob=0.0, max_position_embeddings=512,
                               is_decoder=False, position_embedding_type="absolute")
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
        for i8 in (False, True):
            Q.INT8_STORAGE = i8
            t_mod = timeit(lambda: bq(xb, attention_mask=bmask), n=50)
            print(f"QuantizedBert module fp32, INT8 storage core={i8}: {t_mod:8.1f} us  {Bb * Sb / t_mod:8.1f} M tokens/s")
        try:
            bq(xb, attention_mask=bmask)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                out_b = bq(xb, attention_mask=bmask)
            t_g = timeit(gr.replay, n=50)
            print(f"QuantizedBert module fp32, INT8 storage core, captured HIP graph: {t_g:8.1f} us  {Bb * Sb / t_g:8.1f} M tokens/s  "
                  f"(replay equals eager: {torch.equal(out_b[0], bq(xb, attention_mask=bmask)[0])})")
        except Exception as e:  # noqa: BLE001
            print("capture of the quantised BERT module failed:", type(e).__name__, str(e)[:200])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "int8":
        quantised()
    else:
        main()
