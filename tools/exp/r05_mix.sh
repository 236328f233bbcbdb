#!/bin/bash
# round 5: split8 by v_fma_mix (oeh_common.h: split8_mix, -DOEH_SPLIT_MIX build) against the production split, + the hardware probe
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib
O=gpurun_out/r05_mix
mkdir -p $O
{
tools/probe/mix_probe
for rep in 1 2; do
  echo "## proj_bench production"; python tools/proj_bench.py --no-baseline opt_qkv opt_qkv_novalues bert_qkv | python -c "import sys,json; [print(json.loads(l)['config']['workload'][:40], round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')]"
  echo "## proj_bench split_mix"; OEH_LIB=$L/mix/liboeh_hip.so python tools/proj_bench.py --no-baseline opt_qkv opt_qkv_novalues bert_qkv | python -c "import sys,json; [print(json.loads(l)['config']['workload'][:40], round(json.loads(l)['kernel_us'],2)) for l in sys.stdin if l.startswith('{')]"
done
echo "## attention kernels on fp32 storage: built = production, other = split_mix"
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,dtype=f32,iters=200,ab=$L/mix/liboeh_hip.so" "B=16,H=12,S=512,D=64,causal=1,dtype=f32,int8=1,iters=200,ab=$L/mix/liboeh_hip.so" "B=32,H=12,S=128,D=64,pad=1,dtype=f32,iters=300,ab=$L/mix/liboeh_hip.so" "B=16,H=12,S=512,D=64,causal=1,dtype=f32,clip=1,iters=200,ab=$L/mix/liboeh_hip.so"
echo "## tests on the split_mix build"
OEH_LIB=$L/mix/liboeh_hip.so python -m pytest tests -m gpu -q -x -k "fp32 or f32 or float32 or proj or pairs or triple or int8_modules or quantised" 2>&1 | tail -4
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
