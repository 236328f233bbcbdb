#!/bin/bash
# round 5: the one-pass kernel's tile with the exponentials in two halves, the second one placed between the first half's in-place MFMAs (oeh_attn_flash.inl:
# PIPE_PV), against the -DOEH_NO_PIPE_PV build of the same sources: tests, then same-process A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib/${ALT:-nopipe}/liboeh_hip.so
mkdir -p gpurun_out/r05_pipe_pv
{
python -m pytest tests/test_attn_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,iters=300,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,dtype=bf16,ab=$L" "B=16,H=12,S=512,D=64,iters=300,ab=$L" \
  "B=8,H=12,S=1024,D=64,causal=1,iters=200,ab=$L" "B=4,H=12,S=2048,D=64,causal=1,iters=100,ab=$L" "B=32,H=12,S=256,D=64,causal=1,iters=300,ab=$L" \
  "B=16,H=12,S=512,D=64,causal=1,gmlp=16,iters=300,ab=$L" "B=8,H=16,S=512,D=128,causal=1,iters=200,ab=$L" "B=16,H=12,S=512,D=32,causal=1,iters=300,ab=$L" \
  "B=8,H=12,S=704,D=64,pad=1,iters=200,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,ab=$L"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_pipe_pv/out.txt
