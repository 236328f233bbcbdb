#!/bin/bash
# round 5: the projection GEMM with UNSCALED lo operands (split8_raw, -DOEH_RAW_LO) against the split_mix build
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib
O=gpurun_out/r05_raw
mkdir -p $O
pb() { python tools/proj_bench.py --no-baseline opt_qkv opt_qkv_novalues bert_qkv | python -c "import sys,json; [print(json.loads(l)['config']['workload'][:40], round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')]"; }
{
for rep in 1 2; do
  echo "## proj_bench split_mix (scaled lo)"; OEH_LIB=$L/mix/liboeh_hip.so pb
  echo "## proj_bench raw lo"; OEH_LIB=$L/raw/liboeh_hip.so pb
done
echo "## tests on the raw-lo build"
OEH_LIB=$L/raw/liboeh_hip.so python -m pytest tests/test_proj_gpu.py tests/test_modules_gpu.py -m gpu -q -k "proj or int8_modules or quantised or fuse" 2>&1 | tail -15
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
