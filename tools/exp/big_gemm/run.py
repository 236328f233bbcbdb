"""EXPERIMENT: tools/exp/big_gemm/big_gemm.hip against torch (hipBLASLt) on the projections' shapes."""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libbig_gemm_%s.so" % os.environ.get("BIG_DBG", "0")))
lib.big_gemm_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_void_p]
lib.big_gemm_f16.restype = C.c_int


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


shapes = [(8192, 2304, 768), (8192, 768, 768)] if os.environ.get("BIG_DBG") else [(8192, 2304, 768), (8192, 768, 768), (4096, 2304, 768), (8192, 3072, 768)]
for (M, N, K) in shapes:
    torch.manual_seed(0)
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.float16)
    bias = torch.randn(N, device="cuda")
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)

    def mine():
        rc = lib.big_gemm_f16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, K, K, N, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    bh = bias.half()

    def theirs():
        return torch.nn.functional.linear(a, w, bh)

    mine()
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t() + bias
    err = (c.float() - ref).abs().max().item()
    err_lib = (theirs().float() - ref).abs().max().item()
    t1, t2 = timed(mine), timed(theirs)
    print(f"M={M} N={N} K={K}: big_gemm {t1:.1f} us ({2 * M * N * K / t1 / 1e9:.2f} PFLOP/s), library {t2:.1f} us; max err {err:.3e} (library {err_lib:.3e})", flush=True)

lib.big_gemm_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_float, C.c_void_p]
lib.big_gemm_pairs.restype = C.c_int
lib.big_gemm_pairs_m4.argtypes = lib.big_gemm_pairs.argtypes
lib.big_gemm_pairs_m4.restype = C.c_int
for (M, N, K, fn_) in [(8192, 2304, 768, lib.big_gemm_pairs), (4096, 2304, 768, lib.big_gemm_pairs), (4096, 2304, 768, lib.big_gemm_pairs_m4), (8192, 2304, 768, lib.big_gemm_pairs_m4)]:
    torch.manual_seed(1)
    x = torch.randn(M, K, device="cuda")
    x[:, ::37] *= 30.0
    wi = torch.randint(-128, 128, (N, K), device="cuda").to(torch.float16)
    bias = torch.randn(N, device="cuda") * 0.1
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)
    alpha = 0.003

    def mine2():
        rc = fn_(x.data_ptr(), wi.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, K, K, N, alpha, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    mine2()
    torch.cuda.synchronize()
    ref = (x.double() @ wi.double().t()) * alpha + bias.double()
    err = ((c.double() - ref).abs() / (ref.abs() + 1.0)).max().item()
    t1 = timed(mine2)
    print(f"pairs M={M} N={N} K={K} [{'128 x 288' if fn_ is lib.big_gemm_pairs_m4 else '256 x 288'}]: big_gemm_pairs {t1:.1f} us ({4 * M * N * K / t1 / 1e9:.2f} PFLOP/s executed); max rel err of the fp16 output {err:.3e}", flush=True)
