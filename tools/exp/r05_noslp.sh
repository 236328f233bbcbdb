#!/bin/bash
# round 5: the library built with -fno-slp-vectorize (no compiler-formed v_pk_*_f32) against production, same process
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib/noslp/liboeh_hip.so
mkdir -p gpurun_out/r05_noslp
{
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,iters=300,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,clip=1,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1,ab=$L" \
  "B=16,H=12,S=512,D=64,causal=1,iters=300,i8=1,dtype=f16,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,i8=1,dtype=f32,ab=$L" "B=32,H=12,S=128,D=64,pad=1,iters=400,ab=$L" \
  "B=32,H=12,S=128,D=64,pad=1,gmlp=16,iters=400,ab=$L" "B=32,H=12,S=128,D=64,pad=1,int8=1,iters=400,ab=$L" "B=32,H=12,S=128,D=64,pad=1,i8=1,dtype=f32,iters=400,ab=$L" \
  "B=16,H=12,S=512,D=64,causal=1,dtype=f32,iters=200,ab=$L" "B=16,H=12,S=512,D=64,causal=1,dtype=f32,int8=1,iters=200,ab=$L" "B=8,H=12,S=1024,D=64,causal=1,iters=200,ab=$L" \
  "B=8,H=16,S=512,D=128,causal=1,iters=200,ab=$L" "B=16,H=12,S=512,D=64,iters=300,ab=$L"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_noslp/out.txt
