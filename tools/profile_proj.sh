#!/bin/bash
# rocprofv3 evidence for the projection GEMM (oeh_proj_quant_i8), run through gpurun from the repo root:  tools/profile_proj.sh r04
# (1) bench lines with the same-process library baseline, (2) --kernel-trace --stats of the new kernel alone per workload,
# (3) SQ counters in two --pmc passes (never combined with tracing), FETCH_SIZE / WRITE_SIZE each in its own pass.
set -u
ROUND=$1
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/proj_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/proj_bench.py" > "$OUT/${ROUND}_proj_bench_lines.jsonl" 2> "$OUT/bench.log"
for w in opt_qkv opt_qkv_novalues opt_out_proj bert_qkv; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w/trace" -- python3 "$ROOT/tools/proj_bench.py" --no-baseline --steps 10 $w > "$OUT/$w.trace.json" 2> "$OUT/$w.trace.log"
  f=$(ls "$OUT/$w"/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${ROUND}_proj_${w}_kernel_stats.csv"
done
w=opt_qkv
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$w/$name" -- python3 "$ROOT/tools/proj_bench.py" --no-baseline --steps 2 --warmup 1 --layers 2 $w > "$OUT/$w.$name.log" 2>&1; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 - "$OUT" "$ROUND" <<'PY'
import csv, glob, sys, collections
out, rnd = sys.argv[1], sys.argv[2]
lines = []
for name in ("sq1", "sq2", "fetch", "write"):
    for f in glob.glob(f"{out}/opt_qkv/{name}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "oeh_gemm_kernel" not in k: continue
            for c, v in sorted(acc[k].items()):
                lines.append(f"{k:62s} {c:28s} per dispatch {v / n[(k, c)]:16.1f}   ({n[(k, c)]} dispatches)")
open(f"{out}/{rnd}_proj_pmc_opt_qkv.txt", "w").write("\n".join(lines) + "\n")
PY
for w in opt_qkv opt_qkv_novalues opt_out_proj bert_qkv; do rm -rf "$OUT/$w"; done
