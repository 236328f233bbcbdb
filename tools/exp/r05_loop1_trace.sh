#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/r05_loop1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -- python3 $ROOT/tools/proj_bench.py --no-baseline --steps 10 bert_qkv > $O/a.json 2> $O/a.log
export OEH_DEBUG_HOOKS=1 OEH_GEMM_TILE=2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -- python3 $ROOT/tools/proj_bench.py --no-baseline --steps 10 bert_qkv > $O/b.json 2> $O/b.log
for x in a b; do f=$(ls $O/$x/*/*kernel_stats.csv | head -1); grep oeh "$f" | cut -c1-140; done
rm -rf $O/a $O/b
