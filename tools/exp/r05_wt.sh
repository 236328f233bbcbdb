#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
pb() { python tools/proj_bench.py --no-baseline opt_qkv opt_qkv_novalues opt_out_proj bert_qkv 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['config']['workload'][:44], round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')]"; }
python -m pytest tests/test_proj_gpu.py -m gpu -q 2>&1 | tail -2
for rep in 1 2; do echo "## write-through index pieces"; pb; done
