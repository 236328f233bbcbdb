#!/bin/bash
# round 6: the full-row kernel's K-then-V stream as ONE six-slot ring with up to OEH_RING6 tiles in flight, metered (at most two requests per step)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$ROOT/outeffhop_amd/lib
O=$ROOT/gpurun_out/r06_ab6
mkdir -p $O
cd $ROOT
SPECS=("B=16,H=12,S=512,D=64,causal=1,iters=300,clip=1" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1" "B=16,H=12,S=512,D=64,iters=300,clip=1" "B=32,H=12,S=128,D=64,pad=1,iters=300"
  "B=32,H=12,S=128,D=64,pad=1,int8=1,iters=300" "B=32,H=12,S=128,D=64,pad=1,gmlp=16,iters=300" "B=32,H=12,S=256,D=64,causal=1,clip=1,iters=300" "B=16,H=12,S=384,D=64,causal=1,clip=1,iters=300" "B=16,H=12,S=512,D=64,causal=1,iters=300,int8=1")
{
for v in ${VARIANTS:-r6ring4 r6ring3}; do
  echo "# tests on $v"; OEH_LIB=$L/$v/liboeh_hip.so python -m pytest tests/test_attn_gpu.py tests/test_rows_gpu.py -m gpu -q -x -k "not outlier" 2>&1 | tail -2
  echo "# built = $v, other = the production library (built/other < 1: the variant wins)"
  args=()
  for s in "${SPECS[@]}"; do args+=("$s,ab=$L/liboeh_hip.so"); done
  OEH_LIB=$L/$v/liboeh_hip.so python tools/microbench.py "${args[@]}"
done
} 2>&1 | grep -v amdgpu.ids | tee $O/out.txt
