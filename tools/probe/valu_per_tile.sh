#!/bin/bash
# GPU box: VALU instructions per wave of the one-pass kernel for shapes with different tile counts per wave (fixed
# per-wave cost vs cost per 64-key tile).  Grid sizes differ so that the dispatches can be told apart in the CSV.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/valu_per_tile
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/raw" -- python3 "$ROOT/tools/microbench.py" "causal=0,iters=5" "causal=0,S=1024,B=4,iters=5" "causal=1,iters=5" "causal=0,S=128,B=24,off=256,mq=2,iters=5" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flash" not in r["Kernel_Name"]:
            continue
        acc[(r["Grid_Size"], r["Kernel_Name"][:60])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in sorted(acc.items()):
        waves = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
        line = f"grid {k[0]:>8s} {k[1]}: waves {waves:.0f}  " + "  ".join(f"{c} / wave {sum(v) / len(v) / waves:.1f}" for c, v in sorted(d.items()) if c != "SQ_WAVES")
        print(line); fh.write(line + "\n")
PY
rm -rf "$OUT/raw"
