#!/bin/bash
# instruction-cache counters for one microbench spec (GPU box)
set -u
OUT=$1; SPEC=$2
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/ic" -- python3 "$GRAFT_REPO_ROOT/tools/microbench.py" "$SPEC" > "$OUT/ic.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(f"{sys.argv[1]}/ic/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if "oeh_attn" not in row["Kernel_Name"]: continue
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()): print(f"{k:32s} {s / max(n, 1):16.1f} (n={n})")
PY
