import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from outeffhop_amd import ops, attention
from outeffhop_amd.softmax import SOFTMAX_MAPPING
B,H,S,D=2,12,128,64
q,k,v=(torch.randn(B,S,H*D,device='cuda').half().view(B,S,H,D).permute(0,2,1,3) for _ in range(3))
pad=torch.zeros(B,1,1,S,device='cuda')
def t(fn,n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    dt=(time.perf_counter()-t0)/n*1e6
    torch.cuda.synchronize(); return dt
print("ops.attn_fwd host us/call:", round(t(lambda: ops.attn_fwd(q,k,v,scale_div=8.0,key_pad_mask=pad)),2))
sm=SOFTMAX_MAPPING["softmax1"]
print("attention_core host us/call:", round(t(lambda: attention.attention_core(q,k,v,softmax_fn=sm,scale_div=8.0,attention_mask=pad)),2))
p=ops.PreparedAttn(q,k,v,scale_div=8.0,key_pad_mask=pad)
print("PreparedAttn host us/call:", round(t(lambda: p()),2))
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(2000): ops.attn_fwd(q,k,v,scale_div=8.0,key_pad_mask=pad)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(12)
