#!/bin/bash
# round 5: the projection GEMM's in-kernel fp32 split on 32 x (split8_raw_scaled): hardware probe of the split, parity at every magnitude,
# timing against the two alternatives (alt builds: gscaled = [hi | lo 2^11] against W and W 2^-11, graw = unscaled residual of x itself)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib
O=gpurun_out/r05_prescale
mkdir -p $O
pb() { python tools/proj_bench.py --no-baseline opt_qkv opt_qkv_novalues bert_qkv | python -c "import sys,json; [print(json.loads(l)['config']['workload'][:40], round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')]"; }
{
echo "## tools/probe/mix_probe"
./tools/probe/mix_probe
echo "## tests/test_proj_gpu.py"
python -m pytest tests/test_proj_gpu.py -m gpu -q -s 2>&1 | grep -E "^\|x\||passed|failed|Error|assert" | tail -40
for rep in 1 2 3; do
  echo "## proj_bench production (32 x = hi + lo')"; pb
  echo "## proj_bench scaled lo (W 2^-11 in registers)"; OEH_LIB=$L/gscaled/liboeh_hip.so pb
  echo "## proj_bench raw lo (x = hi + lo')"; OEH_LIB=$L/graw/liboeh_hip.so pb
done
echo "## module + quantised tests"
python -m pytest tests/test_modules_gpu.py -m gpu -q 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
