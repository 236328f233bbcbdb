#!/bin/bash
# round 5: attention kernels' fp32 operand pairs with the unscaled residual (-DOEH_PAIR_RAW) against production: tests + same-process A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib/praw/liboeh_hip.so
mkdir -p gpurun_out/r05_praw
{
OEH_LIB=$L python -m pytest tests -m gpu -q -k "fp32 or f32 or float32 or int8_modules or quantised or stanhop or theory or gate or calibrate or module" 2>&1 | tail -12
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,dtype=f32,iters=200,ab=$L" "B=16,H=12,S=512,D=64,causal=1,dtype=f32,int8=1,iters=200,ab=$L" "B=32,H=12,S=128,D=64,pad=1,dtype=f32,iters=300,ab=$L" "B=16,H=12,S=512,D=64,causal=1,dtype=f32,clip=1,iters=200,ab=$L" "B=32,H=12,S=128,D=64,pad=1,dtype=f32,gmlp=16,iters=300,ab=$L" "B=8,H=12,S=1024,D=64,causal=1,dtype=f32,iters=150,ab=$L"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_praw/out.txt
