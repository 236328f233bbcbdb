import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import outeffhop_amd as oa
from outeffhop_amd import quantization as Q
from outeffhop_amd.opt_attention import OPTAttentionWithExtras
from outeffhop_amd.softmax import SOFTMAX_MAPPING
dev = torch.device("cuda:0")
B, S, E, H = 16, 512, 768, 12
fmin = torch.finfo(torch.float32).min
cfg = oa.get_quant_config(); cfg.act_quant.options = dict(percentile=99.999)
with torch.no_grad():
    org = OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
    qm = oa.QuantizedOPTAttentionWithExtras(org, **{**oa.val_qparams(cfg), "quant_dict": {}}).to(dev).eval()
    qm.set_quant_state(weight_quant=True, act_quant=True)
    mask = torch.full((S, S), fmin, device=dev).triu(1)[None, None].expand(B, 1, S, S).contiguous()
    for _ in range(2):
        qm(torch.randn(B, S, E, device=dev), attention_mask=mask)
    qm.fix_ranges()
    x = torch.randn(B, S, E, device=dev)
    for i in range(4):
        qm(x, attention_mask=mask)
        pl = qm.__dict__.get("_oeh_i8_plan")
        print(i, "plan", None if pl is None else type(pl[1]).__name__, "runs", qm.__dict__.get("_i8_plan_runs", 0), "i8 calls", qm.__dict__.get("_i8_calls", 0))
    pl = qm.__dict__.get("_oeh_i8_plan")
    print("plan error:", qm.__dict__.get("_oeh_i8_plan_error"))
    import inspect
    lins = (qm.q_proj, qm.k_proj, qm.v_proj)
    print("int8_index_ok", qm.out_proj.int8_index_ok(B * S), "index_gemm_ok", qm.out_proj.index_gemm_ok(x), "fit", qm.out_proj._int8_weights_fit(), "x contiguous", x.is_contiguous(), x.data_ptr() % 16)
    print("counters", {k: v for k, v in qm.__dict__.items() if k.startswith("_i") or k.startswith("_f")}, qm.out_proj.__dict__.get("_int8_index_calls"))
    qz = qm.out_proj.weight_quantizer.quantizer
    iw = qz.to_integer_forward(qm.out_proj.weight.detach())
    print("out_proj weight ints", float(iw.min()), float(iw.max()), "signed", qz.signed, "int_min/max", qz.int_min, qz.int_max, "scale", float(qz.scale), "nan", bool(torch.isnan(iw).any()), "dtype", iw.dtype)
    print("fit cache", qm.out_proj.__dict__.get("_int8_fit_cache"))
    if pl is not None:
        p = pl[1]
        st = oa.ops._stream()
        lins = (qm.q_proj, qm.k_proj, qm.v_proj)
        print("xkey", p.xkey == (x.shape, x.stride(), x.dtype, x.device), "stream", p.stream == st.value, "flags", p.flags == Q._I8LayerPlan.state_flags(qm, lins, qm.out_proj))
        bad = [(n_, cur is t_, None if t_ is None else (t_._version, v_)) for d_, n_, t_, v_ in p.watch for cur in [d_.get(n_)] if cur is not t_ or (t_ is not None and t_._version != v_)]
        print("watch mismatches", bad[:6])
    torch.cuda.synchronize()
    import cProfile, pstats
    t0 = time.perf_counter()
    for _ in range(50):
        qm(x, attention_mask=mask)
    torch.cuda.synchronize()
    print("eager us per forward", (time.perf_counter() - t0) / 50 * 1e6)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50):
        qm(x, attention_mask=mask)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
