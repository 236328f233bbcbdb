#!/usr/bin/env python3
"""The experiment GEMM (outeffhop_amd/csrc/oeh_gemm.hip, raw-accumulator form) against the library GEMM the product uses today:
results (pair GEMM vs float64) and launch times, alternating in one process.  GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from outeffhop_amd import ops

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(root, "outeffhop_amd", "lib", "exp", "liboeh_gemm.so"))
lib.oeh_exp_gemm_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p]
lib.oeh_exp_gemm_raw.restype = C.c_int


def run(a, w, c, K, pairs):
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.oeh_exp_gemm_raw(a.data_ptr(), w.data_ptr(), c.data_ptr(), c.shape[0], c.shape[1], K, a.stride(0), w.stride(0), c.stride(0), pairs, st)
    assert rc == 0, rc


def timeit(fn, iters=50, rounds=8):
    res = []
    for r in range(rounds):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            res.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(res))


def main():
    torch.manual_seed(0)
    for (M, N, K) in ((8192, 2304, 768), (4096, 2304, 768), (8192, 768, 768), (1000, 2304, 768)):
        x = torch.randn(M, K, device="cuda")
        x[:, ::97] *= 40.0
        wi = torch.randint(-128, 128, (N, K), device="cuda").to(torch.float16)
        pairs = ops.split_pairs(x)                                   # (M, 2K) fp16
        ww = torch.cat([wi, wi * 2.0 ** -11], dim=1).t().contiguous()   # (2K, N)
        ref64 = x.double() @ wi.double().t()
        c_lib = torch.mm(pairs, ww, out_dtype=torch.float32)
        c = torch.empty(M, N, device="cuda")
        run(pairs, wi, c, K, 1)
        torch.cuda.synchronize()
        sc = ref64.abs().max().item()
        e_new = (c.double() - ref64).abs().max().item() / sc
        e_lib = (c_lib.double() - ref64).abs().max().item() / sc
        # plain fp16 form
        xh = x.half()
        c16 = torch.empty(M, N, device="cuda")
        run(xh, wi, c16, K, 0)
        ref16 = xh.double() @ wi.double().t()
        e16 = (c16.double() - ref16).abs().max().item() / sc
        t_new = timeit(lambda: run(pairs, wi, c, K, 1))
        t_lib = timeit(lambda: torch.mm(pairs, ww, out_dtype=torch.float32))
        t16 = timeit(lambda: run(xh, wi, c16, K, 0))
        wt = wi.t().contiguous()
        t16_lib = timeit(lambda: torch.mm(xh, wt, out_dtype=torch.float32))
        t16_lib_h = timeit(lambda: torch.mm(xh, wt))
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K}: pairs err/max new {e_new:.2e} lib {e_lib:.2e} | fp16 err {e16:.2e} | pairs new {t_new:7.1f} us ({2 * fl / t_new * 1e-9:.2f} PF) "
              f"lib {t_lib:7.1f} us ({2 * fl / t_lib * 1e-9:.2f} PF) | fp16 new {t16:7.1f} us ({fl / t16 * 1e-9:.2f} PF) lib(f32 out) {t16_lib:7.1f} lib(f16 out) {t16_lib_h:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
