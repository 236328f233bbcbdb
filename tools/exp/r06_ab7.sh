#!/bin/bash
# round 6: one-pass kernel, V tiles requested ONE tile ahead (K tiles two, as before): -DOEH_SPLITV
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$ROOT/outeffhop_amd/lib
O=$ROOT/gpurun_out/r06_ab7
mkdir -p $O
cd $ROOT
SPECS=("B=16,H=12,S=512,D=64,causal=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300,dtype=bf16" "B=16,H=12,S=512,D=64,iters=300" "B=8,H=12,S=1024,D=64,causal=1,iters=200"
  "B=4,H=12,S=2048,D=64,causal=1,iters=100" "B=32,H=12,S=256,D=64,causal=1,iters=300" "B=8,H=12,S=704,D=64,pad=1,iters=200" "B=64,H=6,S=197,D=64,iters=300" "B=16,H=12,S=512,D=32,causal=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300")
{
v=${V:-r6splitv}
echo "# tests on $v"; OEH_LIB=$L/$v/liboeh_hip.so python -m pytest tests/test_attn_gpu.py -m gpu -q -x -k "not outlier" 2>&1 | tail -2
echo "# built = $v, other = the production library (built/other < 1: the variant wins)"
args=()
for s in "${SPECS[@]}"; do args+=("$s,ab=$L/liboeh_hip.so"); done
OEH_LIB=$L/$v/liboeh_hip.so python tools/microbench.py "${args[@]}"
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
