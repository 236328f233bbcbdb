#!/bin/bash
# HBM traffic per launch of one microbench spec (GPU box): FETCH_SIZE and WRITE_SIZE in their own rocprofv3 --pmc passes.
# usage: tools/pmc_traffic.sh <outdir> <name> "<microbench spec>"   -> prints  name fetch_kb write_kb bytes
set -u
OUT=$1; NAME=$2; SPEC=$3
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/$NAME-$c" -- python3 "$GRAFT_REPO_ROOT/tools/microbench.py" "$SPEC" > "$OUT/$NAME-$c.log" 2>&1
done
python3 - "$OUT" "$NAME" <<'PY'
import csv, glob, sys
out, name = sys.argv[1:3]
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob(f"{out}/{name}-{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "oeh_" not in row["Kernel_Name"]: continue
            k = row["Kernel_Name"].split("(")[0][:60]
            a = per.setdefault(k, [0.0, 0]); a[0] += float(row["Counter_Value"]); a[1] += 1
    v[c] = {k: s / n for k, (s, n) in per.items()}
tot = 0
for k in v["FETCH_SIZE"]:
    f, w = v["FETCH_SIZE"][k], v["WRITE_SIZE"].get(k, 0.0)
    b = (2 * f + w) * 1024  # gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md)
    tot += b
    print(f"{name} {k}: fetch {f:.1f} KB-units write {w:.1f} KB  -> {b / 1e6:.2f} MB per launch")
print(f"{name} TOTAL_BYTES {int(tot)}")
PY
