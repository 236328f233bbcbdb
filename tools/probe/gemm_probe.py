import torch, time
dev="cuda"
def bench(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
M,K=8192,768
x=torch.randn(M,K,device=dev,dtype=torch.float16)
for N in (2304,768):
    w=torch.randn(N,K,device=dev,dtype=torch.float16); b=torch.randn(N,device=dev,dtype=torch.float16)
    wt=w.t().contiguous()
    for lib in ("default","hipblaslt","hipblas"):
        try:
            if lib!="default": torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib,"n/a",e); continue
        t1=bench(lambda: torch.nn.functional.linear(x,w,b))
        t2=bench(lambda: torch.addmm(b,x,wt))
        t3=bench(lambda: torch.mm(x,wt))
        fl=2*M*N*K
        print(f"N={N} {lib:10s} linear {t1:7.1f} us ({fl/t1/1e6:6.0f} TF)  addmm(x,Wt) {t2:7.1f}  mm {t3:7.1f} ({fl/t3/1e6:6.0f} TF)")
