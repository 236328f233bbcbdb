#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
pb() { python tools/proj_bench.py --no-baseline opt_qkv opt_qkv_novalues bert_qkv 2>/dev/null | python -c "import sys,json; print('   '.join(str(round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')))"; }
echo "# kernel_us: opt_qkv | opt_qkv_novalues | bert_qkv"
for rep in 1 2 3 4; do
  echo -n "requests between the MFMAs (production): "; pb
  echo -n "requests in front of the group         : "; OEH_LIB=$ROOT/outeffhop_amd/lib/dmafront/liboeh_hip.so pb
done
