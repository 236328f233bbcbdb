#!/bin/bash
# round 5: the placed second product in the fp32-storage form of the one-pass kernel against the -DOEH_NO_PIPE32 build: tests, same-process A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib/nop32/liboeh_hip.so
mkdir -p gpurun_out/r05_pipe32
{
python -m pytest tests/test_attn_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,dtype=f32,iters=200,ab=$L" "B=16,H=12,S=512,D=64,dtype=f32,iters=200,ab=$L" "B=8,H=12,S=1024,D=64,causal=1,dtype=f32,iters=150,ab=$L" \
  "B=4,H=12,S=2048,D=64,causal=1,dtype=f32,iters=100,ab=$L" "B=64,H=12,S=197,D=64,dtype=f32,iters=150,ab=$L" "B=16,H=12,S=512,D=32,causal=1,dtype=f32,iters=200,ab=$L" "B=16,H=12,S=512,D=64,causal=1,dtype=f32,iters=200,ab=$L"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_pipe32/out.txt
