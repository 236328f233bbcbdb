#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/:
    python3 tools/profile_summary.py r02 gpurun_out/prof_r02
writes gpurun_out/prof_<round>/summary/: <round>_<workload>_kernel_stats.csv (the --stats table, attention kernels and the
largest others), <round>_pmc_<workload>.txt (per-dispatch means of the SQ counters of the attention kernel),
<round>_bench_lines.jsonl and hbm_traffic.json (bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 counts 64 B per
128-B request for wide coalesced reads - MI355X_MICROARCH.md, HBM section).  Copy that directory's files into profiles/."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    rnd, out = sys.argv[1], sys.argv[2]
    dst = os.path.join(out, "summary")
    os.makedirs(dst, exist_ok=True)
    traffic, detail, lines = {}, {}, []
    for jf in sorted(glob.glob(os.path.join(out, "*.json"))):
        w = os.path.basename(jf)[:-5]
        try:
            rec = json.loads([ln for ln in open(jf) if ln.startswith("{")][-1])
            lines.append(rec)
        except Exception:
            rec = None
        stats = glob.glob(os.path.join(out, w, "trace", "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            rows = list(csv.reader(open(stats[0])))
            keep = [rows[0]] + [r for r in rows[1:] if "oeh" in r[0]] + [r for r in rows[1:] if "oeh" not in r[0]][:3]
            csv.writer(open(os.path.join(dst, f"{rnd}_{w}_kernel_stats.csv"), "w"), quoting=csv.QUOTE_ALL).writerows(keep)
        acc = collections.defaultdict(lambda: [0.0, 0])
        kname = None
        for name in ("sq1", "sq2", "sq4", "fetch", "write"):
            for f in glob.glob(os.path.join(out, w, name, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "oeh_attn" not in row["Kernel_Name"]:
                        continue
                    kname = row["Kernel_Name"].split("(")[0]
                    a = acc[(name, row["Counter_Name"])]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
        if acc:
            with open(os.path.join(dst, f"{rnd}_pmc_{w}.txt"), "w") as fh:
                fh.write(f"# {w}: {kname}; rocprofv3 --pmc, one pass per line group, per-dispatch means (bench.py --steps 3 --warmup 1)\n")
                for (name, cn), (s, n) in sorted(acc.items()):
                    fh.write(f"{name:6s} {cn:28s} per-dispatch mean {s / max(n, 1):16.1f}  (n={n})\n")
            f_kb = acc.get(("fetch", "FETCH_SIZE"), [0, 0])
            w_kb = acc.get(("write", "WRITE_SIZE"), [0, 0])
            if f_kb[1] and w_kb[1]:
                fk, wk = f_kb[0] / f_kb[1], w_kb[0] / w_kb[1]
                traffic[w] = int((2 * fk + wk) * 1024)
                detail[w] = {"kernel": kname, "FETCH_SIZE_KB_per_launch": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
                             "algorithmic_bytes_per_launch": None if rec is None else rec["roofline"]["algorithmic_bytes_per_launch"],
                             "source": f"tools/profile_round.sh {rnd}: FETCH_SIZE and WRITE_SIZE each in its own rocprofv3 --pmc pass over bench.py "
                                       f"--workload {w} --steps 3 --warmup 1; bytes = (2*FETCH + WRITE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)"}
    with open(os.path.join(dst, f"{rnd}_bench_lines.jsonl"), "w") as fh:
        for rec in lines:
            fh.write(json.dumps(rec) + "\n")
    json.dump({**traffic, "_detail": detail}, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    for w, b in traffic.items():
        alg = detail[w]["algorithmic_bytes_per_launch"]
        print(f"{w:20s} HBM bytes/launch {b / 1e6:8.2f} MB  algorithmic {alg / 1e6 if alg else float('nan'):8.2f} MB  ratio {b / alg if alg else float('nan'):.3f}")
    for rec in lines:
        print(f"{rec['config']['variant']:28s} {rec['roofline']['kernel_us']:7.2f} us  frac {rec['roofline']['frac']:.3f}  {rec['value'] / 1e6:7.1f} M tok/s")


if __name__ == "__main__":
    main()
