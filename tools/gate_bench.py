import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from outeffhop_amd import _lib
lib = _lib.load()
H, d, S = 12, 64, 128
for B in (8, 32, 128, 512):
    DT = int(os.environ.get("GATE_DT", "0"))
    hid = [torch.randn(B, S, H*d, device="cuda").half() if DT == 0 else torch.randn(B, S, H*d, device="cuda") for _ in range(8)]
    for m in (16, 0, 64):
        mm = max(m, 1)
        w1 = torch.randn(H, mm, d, device="cuda") * 0.02; b1 = torch.zeros(H, mm, device="cuda")
        w2 = torch.randn(H, mm, device="cuda"); b2 = torch.zeros(H, device="cuda")
        out = torch.empty(B, H, S, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        vp = C.c_void_p
        def call(i):
            hd = hid[i % 8]
            rc = lib.oeh_gate_fwd(vp(hd.data_ptr()), 2 if DT else 0, B, S, H, d, hd.stride(0), hd.stride(1), vp(w1.data_ptr()), vp(b1.data_ptr()), vp(w2.data_ptr()) if m else None, vp(b2.data_ptr()) if m else None, m, 0, 1.0, vp(out.data_ptr()), st)
            assert rc == 0, rc
        for i in range(20): call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200): call(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print(f"B={B:4d} m={m:3d}: {us:8.2f} us   hidden {B*S*H*d*2/1e6:.1f} MB -> {B*S*H*d*2/us/1e3:.0f} GB/s")
