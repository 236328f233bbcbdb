#!/bin/bash
# round 6: the order of the q tiles in the block ids of the full-row and INT8-storage kernels (-DOEH_QT_ORDER=1: heavy / light alternating; =2: 7 5 3 1 then 6 4 2 0)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$ROOT/outeffhop_amd/lib
O=$ROOT/gpurun_out/r06_ab8
mkdir -p $O
cd $ROOT
SPECS=("B=16,H=12,S=512,D=64,causal=1,iters=300,clip=1" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1" "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f32,iters=300" "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f16,iters=300"
  "B=32,H=12,S=256,D=64,causal=1,iters=300,clip=1" "B=16,H=12,S=384,D=64,causal=1,iters=300,clip=1" "B=16,H=12,S=512,D=64,iters=300,int8=1" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1")
for V in r6qo1 r6qo2; do
{
echo "# built = $V, other = the production library (built/other < 1: the variant wins)"
args=()
for s in "${SPECS[@]}"; do args+=("$s,ab=$L/liboeh_hip.so"); done
OEH_LIB=$L/$V/liboeh_hip.so python tools/microbench.py "${args[@]}"
} 2>&1 | grep -v amdgpu.ids | tee $O/out_$V.txt
done
