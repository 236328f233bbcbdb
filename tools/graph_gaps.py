#!/usr/bin/env python3
"""Why does a captured graph of the headline launches replay slower than the eager stream?
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/graph_gaps.py run      (GPU box)
    python3 tools/graph_gaps.py analyse <dir>
`run`: the headline workload (OPT-125m attention core, 12 buffer sets) as (a) 60 x 12 eager launches, (b) 60 replays of one
12-launch graph, (c) 60 replays of a 12-launch graph captured on three forked streams (4 launches each), separated by 50-ms pauses.  `analyse`: per phase the kernel durations, the gaps between consecutive attention kernels inside a
step and the gaps across step boundaries, from the trace's timestamps."""
import ctypes as C
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    import torch

    from outeffhop_amd import ops

    B, H, S, D, L = 16, 12, 512, 64, 12
    g = torch.Generator(device="cuda").manual_seed(0)
    fmin = float(np.finfo(np.float32).min)
    calls = []
    for _ in range(L):
        q = (torch.randn(B, S, H * D, device="cuda", generator=g) * D ** -0.5).half().view(B, S, H, D).permute(0, 2, 1, 3)
        k = torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3)
        v = torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3)
        calls.append(ops.PreparedAttn(q, k, v, causal=True, clamp_min=True, mask_min=fmin))
    import time
    cur = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    for _ in range(50):  # clocks up
        for c in calls:
            c(cur())
    torch.cuda.synchronize()
    N = 60
    time.sleep(0.05)
    for _ in range(N):
        for c in calls:
            c(cur())
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        for c in calls:
            c(cur())
    g1.replay()
    torch.cuda.synchronize()
    time.sleep(0.05)
    for _ in range(N):
        g1.replay()
    torch.cuda.synchronize()
    # three forked streams inside the capture: launches 0-3 / 4-7 / 8-11 have no edges between the groups
    g3 = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream() for _ in range(2)]
    with torch.cuda.graph(g3):
        main = torch.cuda.current_stream()
        for s_ in side:
            s_.wait_stream(main)
        for i, c in enumerate(calls):
            grp = i // 4
            if grp == 0:
                c(cur())
            else:
                with torch.cuda.stream(side[grp - 1]):
                    c(cur())
        for s_ in side:
            main.wait_stream(s_)
    g3.replay()
    torch.cuda.synchronize()
    time.sleep(0.05)
    for _ in range(N):
        g3.replay()
    torch.cuda.synchronize()


def analyse(d):
    import numpy as np

    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # phases are separated by >= 50 ms of idle time; the timed phases are the three segments of 720 (+12 warm-up) launches
    att = [r for r in rows if "oeh_attn" in r[2]]
    segs, cur = [], [att[0]]
    for a_, b_ in zip(att[:-1], att[1:]):
        if b_[0] - a_[1] > 20_000_000:
            segs.append(cur)
            cur = []
        cur.append(b_)
    segs.append(cur)
    segs = [sg for sg in segs if len(sg) >= 700]
    names = ["eager", "graph (one stream)", "graph (three forked streams)"]
    for p, ks in enumerate(segs[-3:]):
        ks = ks[-720:]
        dur = np.array([e - s for s, e, _ in ks]) / 1e3
        st = np.array([s for s, e, _ in ks])
        en = np.array([e for s, e, _ in ks])
        gaps = (st[1:] - en[:-1]) / 1e3
        idx = np.arange(len(gaps))
        inner, cross = gaps[(idx % 12) != 11], gaps[(idx % 12) == 11]
        span = (en[-1] - st[0]) / 1e3 / len(ks)
        print(f"{names[p]:30s} kernels {len(ks):4d}  per-launch span {span:6.2f} us   kernel duration mean {dur.mean():6.2f} (p10 {np.percentile(dur, 10):.2f} p90 {np.percentile(dur, 90):.2f})   "
              f"gap inside a step mean {inner.mean():5.2f} us   gap across steps mean {cross.mean():6.2f} us (max {cross.max():.1f})   overlapped kernels: {int((gaps < 0).sum())}")


def timing():
    """Without a profiler: per-launch time by events for eager launches and for replays of graphs of 12 / 48 / 96 launches (a
    fixed cost per replay shows as a per-launch excess that shrinks with the graph's length), and for two graph objects of
    12 launches replayed alternately."""
    import numpy as np
    import torch

    from outeffhop_amd import ops

    B, H, S, D, L = 16, 12, 512, 64, 12
    g = torch.Generator(device="cuda").manual_seed(0)
    fmin = float(np.finfo(np.float32).min)
    calls = []
    for _ in range(L):
        q = (torch.randn(B, S, H * D, device="cuda", generator=g) * D ** -0.5).half().view(B, S, H, D).permute(0, 2, 1, 3)
        k = torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3)
        v = torch.randn(B, S, H * D, device="cuda", generator=g).half().view(B, S, H, D).permute(0, 2, 1, 3)
        calls.append(ops.PreparedAttn(q, k, v, causal=True, clamp_min=True, mask_min=fmin))
    cur = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731

    def capture(n):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(n):
                calls[i % L](cur())
        gr.replay()
        torch.cuda.synchronize()
        return gr

    graphs = {n: capture(n) for n in (12, 48, 96)}
    g12b = capture(12)

    def timed(fn, launches):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        out = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / launches)
        return float(np.median(out))

    def eager():
        for _ in range(40):
            for c in calls:
                c(cur())

    for _ in range(5):
        eager()
    res = {"eager": timed(eager, 480)}
    for n, gr in graphs.items():
        reps = 480 // n
        res[f"graph of {n}"] = timed(lambda gr=gr, reps=reps: [gr.replay() for _ in range(reps)], n * reps)
    res["two graphs of 12, alternating"] = timed(lambda: [(graphs[12].replay(), g12b.replay()) for _ in range(20)], 480)
    e = res["eager"]
    for k_, v_ in res.items():
        extra = "" if k_ == "eager" else f"   excess per launch {v_ - e:+.2f} us"
        if k_.startswith("graph of"):
            n = int(k_.split()[-1])
            extra += f" = {n * (v_ - e):.1f} us per replay"
        print(f"{k_:32s} {v_:7.2f} us per launch{extra}")


if __name__ == "__main__":
    if sys.argv[1] == "timing":
        timing()
    elif sys.argv[1] == "run":
        run()
    else:
        analyse(sys.argv[2])
