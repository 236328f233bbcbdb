import torch, time
torch.manual_seed(0)
M,N,K=8192,768,768
x=torch.randn(M,K,device='cuda')
iw=torch.randint(-127,128,(N,K),device='cuda').float()
sw=0.003
w=iw*sw
b=torch.randn(N,device='cuda')
def timeit(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
ref=torch.nn.functional.linear(x.double(),w.double(),b.double())
y32=torch.nn.functional.linear(x,w,b)
iw16=iw.half().t().contiguous()   # (K,N)
def split(x):
    hi=x.half(); lo=((x-hi.float())*2048.0).half(); return hi,lo
def lin_split(x):
    hi,lo=split(x)
    a=torch.mm(hi,iw16,out_dtype=torch.float32)
    c=torch.mm(lo,iw16,out_dtype=torch.float32)
    return (a+c*(1/2048.0))*sw+b
try:
    y=lin_split(x)
    print("err fp32 lib", float((y32.double()-ref).abs().max()), "err split", float((y.double()-ref).abs().max()), "ref max", float(ref.abs().max()))
    print("t fp32 linear", timeit(lambda: torch.nn.functional.linear(x,w,b)), "t split", timeit(lambda: lin_split(x)))
    hi,lo=split(x)
    print("t one mm f16->f32", timeit(lambda: torch.mm(hi,iw16,out_dtype=torch.float32)), "t split only", timeit(lambda: split(x)))
    xx=torch.cat([hi,lo],dim=1); ww=torch.cat([iw16, iw16*(1/2048.0)],dim=0)
    print("t cat-K mm", timeit(lambda: torch.mm(xx,ww,out_dtype=torch.float32)), "err", float(((torch.mm(xx,ww,out_dtype=torch.float32)*sw+b).double()-ref).abs().max()))
except Exception as e:
    print("FAILED", repr(e))
