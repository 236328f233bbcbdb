#!/bin/bash
# round 6: s_setprio 2 for the K / V (matrix-core, memory-paced) phases of the full-row and INT8-storage kernels, 0 for their vector chain (-DOEH_PHASE_PRIO)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$ROOT/outeffhop_amd/lib
O=$ROOT/gpurun_out/r06_ab5
mkdir -p $O
cd $ROOT
SPECS=("B=16,H=12,S=512,D=64,causal=1,iters=300,clip=1" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1" "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f32,iters=300" "B=16,H=12,S=512,D=64,causal=1,i8=1,dtype=f16,iters=300"
  "B=32,H=12,S=128,D=64,pad=1,iters=300" "B=32,H=12,S=128,D=64,pad=1,int8=1,iters=300" "B=32,H=12,S=128,D=64,pad=1,i8=1,dtype=f32,iters=300" "B=16,H=12,S=512,D=64,iters=300,int8=1" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1")
{
echo "# built = ${V:-r6pp}, other = the production library (built/other < 1: the variant wins)"
args=()
for s in "${SPECS[@]}"; do args+=("$s,ab=$L/liboeh_hip.so"); done
OEH_LIB=$L/${V:-r6pp}/liboeh_hip.so python tools/microbench.py "${args[@]}"
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
