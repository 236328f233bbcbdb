#!/bin/bash
# round 5: the full-row and integer kernels' units built with -fno-slp-vectorize (make FULLROW_FLAGS=-fno-slp-vectorize: no compiler-formed v_pk_*_f32) = "other",
# against production, same process, two rounds
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib/frns/liboeh_hip.so
mkdir -p gpurun_out/r05_noslp
{
OEH_LIB=$L python -m pytest tests/test_attn_gpu.py -m gpu -q 2>&1 | tail -2
for rep in 1 2; do
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,iters=300,clip=1,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,clip=1,int8=1,ab=$L" \
  "B=16,H=12,S=512,D=64,causal=1,iters=300,i8=1,dtype=f16,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,i8=1,dtype=f32,ab=$L" "B=32,H=12,S=128,D=64,pad=1,iters=400,ab=$L" \
  "B=32,H=12,S=128,D=64,pad=1,gmlp=16,iters=400,ab=$L" "B=32,H=12,S=128,D=64,pad=1,int8=1,iters=400,ab=$L" "B=32,H=12,S=128,D=64,pad=1,i8=1,dtype=f32,iters=400,ab=$L" \
  "B=16,H=12,S=512,D=64,causal=1,dtype=f32,int8=1,iters=200,ab=$L" "B=32,H=12,S=128,D=64,pad=1,dtype=f32,iters=300,ab=$L" "B=32,H=12,S=256,D=64,causal=1,clip=1,iters=300,ab=$L" "B=64,H=12,S=197,D=64,clip=1,iters=200,ab=$L"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_noslp/out.txt
