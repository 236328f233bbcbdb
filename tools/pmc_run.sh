#!/bin/bash
# rocprofv3 PMC passes over the micro-benchmark (GPU box).  usage: tools/pmc_run.sh <outdir> "<microbench spec>"
# Counters are collected in their own passes (never with tracing), per /opt/skills/guides/MI355X_MICROARCH.md.
set -u
OUT=$1; SPEC=$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/tools/microbench.py" "$SPEC" > "$OUT/$name.log" 2>&1
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
pass sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL
pass sq4 SQ_INSTS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("sq1", "sq2", "sq3", "sq4", "fetch", "write"):
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "oeh_attn" not in row["Kernel_Name"]:
                continue
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (s, n) in sorted(acc.items()):
            print(f"{name:6s} {k:28s} per-dispatch mean {s / max(n,1):16.1f}  (n={n})")
PY
