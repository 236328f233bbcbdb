#!/bin/bash
# GPU box (round 6): vector / scalar / transcendental instructions PER WAVE of the two cfg4 kernels (fused INT8 chain on fp16 storage:
# oeh_attn_fast_kernel<32,...,FQ=1>; INT8 storage: oeh_attn_i8_kernel<32,...>) for rows of 5 / 6 / 7 / 8 key tiles (Sk = 320 ... 512, non-causal, every wave the
# same work): the slope is the cost of a 64-key tile (16 elements per lane), the intercept the fixed cost of a wave (prologue + epilogue).  Grid sizes
# differ (B = 4 ... 7) so that the dispatches can be told apart in the CSV.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/valu_per_tile_cfg4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kind in int8 i8; do
  extra=""; [ $kind = i8 ] && extra=",dtype=f32"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d "$OUT/raw_$kind" -- python3 "$ROOT/tools/microbench.py" \
    "B=4,S=512,Sk=320,$kind=1,iters=5$extra" "B=5,S=512,Sk=384,$kind=1,iters=5$extra" "B=6,S=512,Sk=448,$kind=1,iters=5$extra" "B=7,S=512,Sk=512,$kind=1,iters=5$extra" > "$OUT/log_$kind.txt" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/summary.txt", "w") as fh:
    for kind in ("int8", "i8"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(out + f"/raw_{kind}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "oeh_attn" not in r["Kernel_Name"]:
                    continue
                acc[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        pts = {}
        for g, d in sorted(acc.items()):
            waves = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
            tiles = {4: 5, 5: 6, 6: 7, 7: 8}.get(g // (12 * 8 * 256), None)
            per = {c: sum(v) / len(v) / waves for c, v in d.items() if c != "SQ_WAVES"}
            pts[tiles] = per
            line = f"{kind:5s} grid {g:8d} (tiles per wave {tiles}): waves {waves:.0f}  " + "  ".join(f"{c[9:]} {x:.1f}" for c, x in sorted(per.items()))
            print(line); fh.write(line + "\n")
        if 5 in pts and 8 in pts:
            for c in sorted(pts[5]):
                slope = (pts[8][c] - pts[5][c]) / 3.0
                line = f"{kind:5s}   {c[9:]:16s} per 64-key tile {slope:7.1f}   fixed per wave {pts[8][c] - 8 * slope:7.1f}"
                print(line); fh.write(line + "\n")
PY
rm -rf "$OUT"/raw_*
