#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
export OEH_LIB=$ROOT/outeffhop_amd/lib/experiment/liboeh_hip.so OEH_DEBUG_HOOKS=1
pb() { python tools/proj_bench.py --no-baseline $1 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['config']['workload'][:30], round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')]"; }
for d in 0 1 32; do
  echo "## LOOP == 1, OEH_GEMM_DBG=$d (1: no epilogue; 32: no image -> global stores)"; OEH_GEMM_DBG=$d pb bert_qkv
  echo "## opt_qkv_novalues 128 x 288 two per CU, OEH_GEMM_DBG=$d"; OEH_GEMM_DBG=$d pb opt_qkv_novalues
done
