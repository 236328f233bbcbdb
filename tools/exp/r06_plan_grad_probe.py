"""GPU probe (round 6): what the quantised OPT module's FULL path does when autograd is recording, with and without a prebuilt plan."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import outeffhop_amd as oa
from outeffhop_amd import quantization as Q

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T, E, H = 3, 128, 256, 4
cfg = oa.get_quant_config(); cfg.act_quant.options = dict(percentile=99.999)
qp = {**oa.val_qparams(cfg), "quant_dict": {}}
org = oa.OPTAttentionWithExtras(E, H, is_decoder=True, softmax_fn=oa.SOFTMAX_MAPPING["softmax1"]).to(dev).eval()
qm = oa.QuantizedOPTAttentionWithExtras(org, **qp).to(dev).eval()
qm.set_quant_state(weight_quant=True, act_quant=True)
fmin = torch.finfo(torch.float32).min
mask = torch.full((T, T), fmin, device=dev).triu(1)[None, None].expand(B, 1, T, T).contiguous()
with torch.no_grad():
    for _ in range(3):
        qm(torch.randn(B, T, E, device=dev), attention_mask=mask)
    qm.fix_ranges()
    x = torch.randn(B, T, E, device=dev)
    qm(x, attention_mask=mask); qm(x, attention_mask=mask)
print("plan runs after two no_grad forwards:", qm.__dict__.get("_i8_plan_runs", 0))
for plan in (True, False):
    Q.I8_PLAN = plan
    for what, xin in (("params require grad", x), ("input requires grad", x.clone().requires_grad_(True))):
        before = qm.__dict__.get("_i8_plan_runs", 0)
        with torch.enable_grad():
            try:
                out = qm(xin, attention_mask=mask)[0]
                res = f"returned: requires_grad={out.requires_grad} grad_fn={'yes' if out.grad_fn is not None else 'NONE'}"
            except Exception as e:  # noqa: BLE001
                res = f"raised {type(e).__name__}: {str(e)[:90]}"
        print(f"I8_PLAN={plan} {what}: {res}; plan runs +{qm.__dict__.get('_i8_plan_runs', 0) - before}; i8 calls {qm.__dict__.get('_i8_calls', 0)}")
