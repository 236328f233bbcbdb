import os, sys
os.environ["OEH_DEBUG_HOOKS"]="1"
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from outeffhop_amd import _lib, ops
lib=_lib.load()
GEN=(1<<1)|(1<<2)|(1<<3)|(1<<5)|(1<<6)|(1<<7)
fmin=float(np.finfo(np.float32).min)
dev='cuda'
for base in (0,1):
  for padmode in ("right","none"):
    worst=0
    for seed in range(60):
        g=torch.Generator(device=dev).manual_seed(seed)
        B,H,Sq,Sk,D=1,4,283,161,64
        q=torch.randn(B,H,Sq,D,device=dev,generator=g)*D**-0.5; k=torch.randn(B,H,Sk,D,device=dev,generator=g); v=torch.randn(B,H,Sk,D,device=dev,generator=g)
        kw=dict(softmax=ops.SoftmaxSpec(base), scale=1.0, mask_min=fmin)
        if padmode=="right":
            pad=torch.zeros(B,Sk,device=dev); L=int(torch.randint(1,Sk,(1,)).item()); pad[0,L:]=fmin; kw["key_pad_mask"]=pad
        lib.oeh_debug_set_variant(0,0); a=ops.attn_fwd(q,k,v,**kw)
        lib.oeh_debug_set_variant(GEN,0); r=ops.attn_fwd(q,k,v,**kw); lib.oeh_debug_set_variant(0,0)
        worst=max(worst,float((a-r).abs().max()))
    print("base",base,"pad",padmode,"variant",ops.attn_variant(B,H,Sq,Sk,D,torch.float32,base=base,key_pad=padmode=="right",mask_min=fmin),"worst abs err vs any-shape kernel",f"{worst:.3e}")
