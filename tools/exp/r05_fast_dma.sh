#!/bin/bash
# round 5: full-row kernel with the scalar-base LDS-DMA stream + the INT8-storage core's in-place quantise pass, against the previous build (lib/prev), one process
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
P=$ROOT/outeffhop_amd/lib/prev/liboeh_hip.so
mkdir -p gpurun_out/r05_fast
H="B=16,H=12,S=512,D=64,causal=1,iters=300"
{
python tools/microbench.py "$H,clip=1,ab=$P" "$H,int8=1,ab=$P" "$H,i8=1,dtype=f32,ab=$P" "$H,i8=1,dtype=f16,ab=$P" "B=32,H=12,S=128,D=64,pad=1,iters=400,ab=$P" "B=32,H=12,S=128,D=64,pad=1,gmlp=16,iters=400,ab=$P" \
  "B=32,H=12,S=128,D=64,pad=1,int8=1,iters=400,ab=$P" "B=32,H=12,S=128,D=64,pad=1,i8=1,dtype=f32,iters=400,ab=$P" "B=32,H=12,S=256,D=64,causal=1,clip=1,iters=300,ab=$P" "B=24,H=12,S=320,D=64,causal=1,iters=300,ab=$P" \
  "B=16,H=12,S=512,D=64,causal=1,clip=1,int8=1,iters=300,ab=$P" "B=8,H=16,S=512,D=128,causal=1,clip=1,iters=200,off=2048,ab=$P" "B=64,H=12,S=128,D=32,pad=1,iters=300,ab=$P"
python -m pytest tests/test_attn_gpu.py tests/test_modules_gpu.py -m gpu -q -x 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_fast/ab.txt
