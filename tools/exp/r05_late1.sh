#!/bin/bash
# round 5: the one-pass kernel with stage 1 (K1, V1) requested after the wait for Q + K0 instead of with them (alt build -DOEH_LATE_STAGE1 = "other")
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/outeffhop_amd/lib/late1/liboeh_hip.so
mkdir -p gpurun_out/r05_late1
{
OEH_LIB=$L python -m pytest tests/test_attn_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/microbench.py "B=16,H=12,S=512,D=64,causal=1,iters=300,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,dtype=bf16,ab=$L" "B=16,H=12,S=512,D=64,iters=300,ab=$L" \
  "B=8,H=12,S=1024,D=64,causal=1,iters=200,ab=$L" "B=4,H=12,S=2048,D=64,causal=1,iters=100,ab=$L" "B=32,H=12,S=256,D=64,causal=1,iters=300,ab=$L" \
  "B=64,H=12,S=197,D=64,iters=200,ab=$L" "B=16,H=12,S=512,D=32,causal=1,iters=300,ab=$L" "B=8,H=12,S=704,D=64,pad=1,iters=200,ab=$L" "B=16,H=12,S=512,D=64,causal=1,iters=300,ab=$L"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_late1/out.txt
