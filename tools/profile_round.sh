#!/bin/bash
# One round's rocprofv3 evidence, collected on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02 [workload ...]
# Per workload of bench.py: (1) --kernel-trace --stats summary, (2) SQ counters in four --pmc passes, (3) FETCH_SIZE and
# WRITE_SIZE each in its own --pmc pass (never combined with tracing: /opt/skills/guides/MI355X_MICROARCH.md).  Raw output
# goes under gpurun_out/prof_<round>/ (scratch); tools/profile_summary.py turns it into the tracked files under profiles/.
set -u
ROUND=$1; shift
WLS=${@:-opt_softmax1 opt_clipped opt_int8 opt_int8_i8 opt_int8_i8_o8 opt_softmax1_fp32 opt_int8_fp32 bert_softmax1 bert_gated bert_int8 bert_int8_i8 bert_softmax1_fp32 bert_gated_fp32 stanhop}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in $WLS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w/trace" -- python3 "$ROOT/bench.py" --workload $w --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/$w.json" 2> "$OUT/$w.trace.log"
  pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$w/$name" -- python3 "$ROOT/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/$w.$name.log" 2>&1; }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  case $w in opt_softmax1|opt_int8|opt_int8_i8|opt_int8_i8_o8|opt_int8_fp32|opt_clipped|bert_int8_i8)
    pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
    pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
    pass sq4 SQ_INSTS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH
  esac
done
python3 "$ROOT/tools/profile_summary.py" "$ROUND" "$OUT"
# keep only the summary (gpurun copies back at most 64 MiB)
for w in $WLS; do rm -rf "$OUT/$w"; done
